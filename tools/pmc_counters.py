#!/usr/bin/env python3
"""profiles/pmc_counters.json and profiles/pmc_traffic.json from one round's collection (tools/collect_round.sh <tag>):
   usage: python tools/pmc_counters.py gpurun_out/<tag> <tag>
Per workload: the render kernel's SQ_INSTS_VALU_FLOPS_FP32 (wave instructions per launch), its lane utilisation
SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) and the rays of that launch (the bench line of the same collection) - what
bench.py's roofline.counter_frac is computed from - and the memory-side bytes per launch (2 x FETCH_SIZE + WRITE_SIZE KiB)."""
import json
import os
import sys

out_dir, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
counters, traffic, valu = {}, {}, {}
for w in ("weekend-1080p-256spp", "dreadnaught-1080p-128spp", "meshdemo-4k-64spp", "dootdoot-4k-64spp", "meshdemo-4k-1024spp"):
    if not os.path.exists(os.path.join(out_dir, f"pmc_{w}.json")):
        continue                                                              # (the full C4 frame has SQ / TA passes only, and not in every collection)
    pmc = json.load(open(os.path.join(out_dir, f"pmc_{w}.json")))
    line = json.loads(open(os.path.join(out_dir, f"bench_{w}.json")).read().strip().splitlines()[-1])
    rays = sum(line["rays_per_step"][k] for k in ("primary", "secondary", "shadow")) * line["steps"] // max(line["roofline"]["launches"], 1)      # a frame beyond the segment-sum cap takes several launches
    lane_util = pmc["SQ_THREAD_CYCLES_VALU"] / (64.0 * pmc["SQ_ACTIVE_INST_VALU"])
    counters[w] = {"valu_flops_fp32_wave_instr": pmc["SQ_INSTS_VALU_FLOPS_FP32"], "lane_util": round(lane_util, 4), "rays_per_launch": rays,
                   "valu_insts": pmc["SQ_INSTS_VALU"], "source": f"profiles/{tag}_pmc_{w}.json"}
    if "FETCH_SIZE" in pmc:
        traffic[w] = int((2 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024)
    valu[w] = pmc["SQ_INSTS_VALU"]
counters["_note"] = ("per render_kernel launch (one launch = one frame of the named workload); counter_frac = valu_flops_fp32_wave_instr x 64 x lane_util / rays_per_launch "
                     "x the measured rays per second / 157.3 TFLOP/s (bench.py)")
traffic["valu_insts"] = valu
traffic["_note"] = (f"per render_kernel launch, from profiles/{tag}_pmc_<workload>.json: bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950 halves FETCH_SIZE; Infinity-Cache hits are "
                    "included, so an upper bound on HBM bytes; the algorithmic part is one float3 segment sum per pixel and 8 samples, the rest is write-back of the per-lane private "
                    "segment: the light colour parked during every shadow ray, the deferred-ray list); valu_insts = SQ_INSTS_VALU")
json.dump(counters, open(os.path.join(root, "profiles", "pmc_counters.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(counters, indent=1))
