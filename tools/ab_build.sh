#!/bin/bash
# A/B helper: a second copy of the tree at <commit> under _ab_head/ (git archive: the index is not touched), library built.
# On the GPU box: python _ab_head/tools/<tool>.py next to python tools/<tool>.py, same box, alternating.
set -e
cd "$(dirname "$0")/.."
rm -rf _ab_head && mkdir _ab_head
git archive "${1:-HEAD}" | tar -x -C _ab_head
rm -rf _ab_head/tests/golden _ab_head/profiles
make -s -j8 -C _ab_head/coldrec_amd/csrc > /dev/null
rm -rf _ab_head/coldrec_amd/lib/obj
echo "built _ab_head at $(git rev-parse --short "${1:-HEAD}")"
