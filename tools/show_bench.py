import json,sys
for f in sys.argv[1:]:
    j=json.loads(open(f).read().strip().splitlines()[-1])
    r=j["roofline"]
    print(f, j["value"], j["ms_per_step"], r["kernel"], r["frac"], r["algorithmic_bytes_per_launch"], r["avg_launch_ms"], j["roofline_valu"].get("insts_per_kmer"), j.get("parity_vs_oracle"), (j["cpu_baseline"] or {}).get("value"))
