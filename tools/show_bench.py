import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def show(k,e):
    print(k, e.get("value"), e.get("ms_per_step"), "cpu", e.get("cpu_baseline",{}).get("value"), "x", e.get("gpu_over_cpu"), (e.get("cpu_baseline",{}).get("sample") or "")[:64])
show("C2",d); show("C3",d["target_scene"])
for k,e in d["configs"].items():
    if "value" in e: show(k,e)
