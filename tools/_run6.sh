set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05e; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_score_topk_gpu.py -x -q -k "f16 or dma or workgroup or lockstep or seeded" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
timeout 600 python -m pytest tests/test_g16_user_cold_gpu.py tests/test_g17_ngcf_plugin_gpu.py tests/test_train_gpu.py -x -q -s > $OUT/tests_g16.log 2>&1; grep -E "^g1[67]|passed|failed|Error|error" $OUT/tests_g16.log | tail -12
./tools/probes/mfma_energy_probe 2>&1 | grep "4 waves, 4 acc" | tail -2 > $OUT/bare.log; cat $OUT/bare.log
timeout 300 python tools/f16_ab.py --rounds 4 --arms "0/noseed,2/noseed,0,2" 2>&1 | grep -v amdgpu.ids > $OUT/f16_ab.log; cat $OUT/f16_ab.log
