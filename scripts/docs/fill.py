"""python scripts/docs/fill.py: DESIGN.md / README.md from their templates, the @FIGURES@ taken from profiles/r05_*_bench_line.json,
profiles/r05_shard_probe.txt and gpurun_out/r05/c4_one_gpu.json (after scripts/round_profiles.sh; edit the TEMPLATES, not the outputs)."""
import json,re
def line(w): return json.load(open(f"profiles/r05_{w}_bench_line.json"))
c2,c1,c3,c5=line("c2"),line("c1"),line("c3"),line("c5")
sp={}
for l in open("profiles/r05_shard_probe.txt"):
    t=l.split()
    if t: sp[t[0]]=float(t[1])
try:
    c4=json.load(open("gpurun_out/r05/c4_one_gpu.json"))
except OSError:  # (the probe's own line in profiles/ carries the same figure: "... 14361 M evals/s ...")
    c4={"value": float(re.search(r"^c4_one_gpu.*?([0-9.]+) M evals/s ", open("profiles/r05_shard_probe.txt").read(), re.M).group(1))}
S1,S2,S4,S8=sp["c4_one_gpu"],sp["c4_strong2_rank0"],sp["c4_strong4_rank0"],sp["c4_strong8_rank0"]
v={
 "C2V":"%.2f"%(c2["value"]/1e3),"C2MS":"%.2f"%c2["ms_per_step"],"C2K":"%.2f"%c2["roofline"]["kernel_avg_ms"],"C2FRAC":"%.2f"%c2["roofline"]["frac"],
 "C2ISO":"%.2f"%c2["roofline"]["kernel_isolated_ms"],"C2T":"%.1f"%(c2["roofline"]["traffic"]/1e9),"C2UP":"%.2f"%c2["upload_inclusive"]["ms_per_step"],
 "C3V":"%.1f"%(c3["value"]/1e3),"C3MS":"%.1f"%c3["ms_per_step"],"C3K":"%.1f"%c3["roofline"]["kernel_avg_ms"],
 "C1V":"%.2f"%(c1["value"]/1e3),"C1MS":"%.2f"%c1["ms_per_step"],"C1K":"%.2f"%c1["roofline"]["kernel_avg_ms"],
 "C5V":"%.1f"%(c5["value"]/1e3),"C5MS":"%.2f"%c5["ms_per_step"],
 "C4MS":"%.1f"%S1,"C4V":"%.1f"%(c4["value"]/1e3),"C4R":"%.2f"%S8,"CPUV":"%.1f"%c2["cpu_baseline"]["value"],
 "S1":"%.1f"%S1,"S2":"%.1f"%S2,"S4":"%.1f"%S4,"S8":"%.2f"%S8,"E2":"%.2f"%(S1/(2*S2)),"E4":"%.2f"%(S1/(4*S4)),"E8":"%.2f"%(S1/(8*S8)),
}
for src,dst in (("scripts/docs/DESIGN.tmpl","DESIGN.md"),("scripts/docs/README.tmpl","README.md")):
    s=open(src).read()
    for k,x in v.items(): s=s.replace("@%s@"%k,x)
    left=re.findall(r"@[A-Z0-9]+@",s)
    print(dst,"unfilled:",left)
    open(dst,"w").write(s)
print(v)
