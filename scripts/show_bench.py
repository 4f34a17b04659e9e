import sys,json
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): 
        if line: print(line[:300])
        continue
    d=json.loads(line)
    print(sys.argv[1], "ms/step", round(d["ms_per_step"],2), "value %.3g"%d["value"], "kept", d["kept_per_step"], d["config"]["parallelism"][:40], d.get("roofline",{}).get("kernels_ms"), d["host_split_s"])
