set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05d; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_score_topk_gpu.py -x -q -k "f16 or dma or workgroup or lockstep or seeded" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
./tools/probes/mfma_energy_probe 2>&1 | grep "4 waves, 4 acc" | tail -2 > $OUT/bare.log; cat $OUT/bare.log
timeout 300 python tools/f16_ab.py --rounds 4 2>&1 | grep -v amdgpu.ids > $OUT/f16_ab.log; cat $OUT/f16_ab.log
PLIB=$PWD/coldrec_amd/lib/libcoldrec_hip_profile.so
for abl in 0 1 5; do
  CRH_LIB=$PLIB CRH_SCORE_ABLATE=$abl timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d $OUT/pmc_$abl -- python3 tools/f16_ab.py --rounds 1 > $OUT/pmc_$abl.out 2> $OUT/pmc_$abl.err
  python3 tools/prof_summary.py r05d_abl$abl $OUT/pmc_$abl $OUT/pmc_$abl > /dev/null 2>&1
  rm -rf $OUT/pmc_$abl
done
mkdir -p gpurun_out/profiles_r05d; cp profiles/r05d_* gpurun_out/profiles_r05d/
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('profiles/r05d_abl*_pmc.json')):
    d=json.load(open(f))
    for k,v in d.items():
        if 'score_topk' in k and 'pack' not in k and v.get('_duration_ns',0)>1e8:
            cyc=v['GRBM_GUI_ACTIVE']/8; busy=v['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*cyc); ghz=cyc/v['_duration_ns']
            wc=v['SQ_WAVE_CYCLES']
            print(f.split('/')[-1][:12], k[20:60], 'ms %.1f busy %.3f clock %.3f GHz busyGHz %.3f wait_any %.3f wait_inst %.3f active %.3f lds %.3f'%(v['_duration_ns']/1e6,busy,ghz,busy*ghz,v['SQ_WAIT_ANY']/wc,v['SQ_WAIT_INST_ANY']/wc,v['SQ_ACTIVE_INST_ANY']/wc,v['SQ_WAIT_INST_LDS']/wc))
PY
