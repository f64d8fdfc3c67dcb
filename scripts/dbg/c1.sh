python -m pytest tests/test_parity_vpm_gpu.py tests/test_rotated_gpu.py tests/test_exact_pass_gpu.py -k "vpm" -x -q 2>&1 | tail -3
for r in 1 2 3; do for v in r4 head default; do
  if [ $v == default ]; then unset GVPM_HIP_LIB; else export GVPM_HIP_LIB=$PWD/build/variants/libgvpm_hip_$v.so; fi
  echo -n "$r $v c1: "; python bench.py --workload c1 --steps 16 --warmup 2 --no-cpu-baseline --no-parity --no-upload-inclusive --no-isolated 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%.0f Mev/s step %.3f ms kernel %.3f" % (d["value"], d["ms_per_step"], r["kernel_avg_ms"]))'
done; done
