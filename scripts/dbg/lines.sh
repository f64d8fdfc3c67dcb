for v in r4 default; do
  if [ $v == default ]; then unset GVPM_HIP_LIB; else export GVPM_HIP_LIB=$PWD/build/variants/libgvpm_hip_$v.so; fi
  for wl in c1 c3 c5; do
    st=16; [ $wl == c3 ] && st=8
    echo -n "$v $wl: "; python bench.py --workload $wl --steps $st --warmup 2 --no-cpu-baseline --no-parity --no-upload-inclusive --no-isolated 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%.0f Mev/s step %.3f ms kernel %.3f frac %.3f" % (d["value"], d["ms_per_step"], r["kernel_avg_ms"], r["frac"]))'
  done
done
