import json,sys
f=sys.argv[1]
try:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    p=d.get("parity",{}); q=d.get("parity_trained",{}); r=d.get("roofline",{})
    print(f, d["dtype"], d["value"], d["ms_per_step"], "parity logit/heat", p.get("logit_max_rel"), p.get("heatmap_max_rel"), "trained", q.get("logit_max_rel"), q.get("heatmap_max_rel"), "roof", r.get("avg_us"), r.get("frac"), "loss", d["config"]["final_loss"])
except Exception as e:
    print(f, "ERR", e)
