for pr in "0,1,-1" "0,-1,-1" "1,-1,-1" "0,-1,0" "0,0,0"; do for w in 8 12; do
echo -n "prio $pr waves $w: "; GVPM_STREAM_PRIORITIES=$pr GVPM_WAVES_PER_CU=$w python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-parity --no-upload-inclusive --no-isolated 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%.0f Mev/s step %.3f ms eval %.2f trav %.2f build %.2f" % (d["value"], d["ms_per_step"], r["kernel_avg_ms"], r["traverse_avg_ms"], r["build_avg_ms"]))'
done; done
