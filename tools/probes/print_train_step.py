import json,sys
d=json.loads(sys.stdin.readlines()[-1]); t=d["train_step"]; f=t.get("fp16_blocks") or {}
print("eval", d["value"], "| fp32 step", t["ms_per_step"], "eager", t["hip_graph"]["eager_ms_per_step"], "| fp16 step", f.get("ms_per_step"), "eager", (f.get("hip_graph") or {}).get("eager_ms_per_step"))
e = d.get("fp16_blocks_eval") or {}
print("fp16 eval", e.get("value") or e.get("images_per_s"), str(e)[:200])
