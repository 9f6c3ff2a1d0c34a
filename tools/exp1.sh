set -e
echo "== baseline"; timeout -k 10 120 python tools/sweep_bench.py --ablate 0 --panel-rows 524288
echo "== no-scatter ids (hot ids contiguous)"; ADMM_SYNTH_NOSCATTER=1 timeout -k 10 120 python tools/sweep_bench.py --ablate 0 --panel-rows 524288
echo "== u32 atomics"; cp admm_amd/libadmm_hip.so /tmp/keep.so; cp tools/libadmm_hip_u32.so admm_amd/libadmm_hip.so; timeout -k 10 120 python tools/sweep_bench.py --ablate 0 --panel-rows 524288; cp /tmp/keep.so admm_amd/libadmm_hip.so
