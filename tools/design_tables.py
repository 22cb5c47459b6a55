"""Development tool: the counter and throughput tables of the design notes from the records under profiles/.
usage: python tools/design_tables.py [tag]      (default r03)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"


def load(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.load(f)


names = {"C2": "C2 RTIOW", "C3": "C3 textured", "C4": "C4 Cornell (no tree)", "C5": "C5 Next-Week-like (28-byte nodes in LDS)"}
print("| config | VALU issue busy | lanes per instruction | roofline.frac | waiting share of wave cycles | global loads per path | HBM B per path | vector instructions per path (wave level) |")
print("|---|---|---|---|---|---|---|---|")
for c in ("C2", "C3", "C4", "C5"):
    d = load(f"{tag}_pmc_{c}.json")
    busy, lanes = d["valu_issue_busy_share_of_simd_cycles"], d["lanes_active_per_valu_instruction"]
    print(f"| {names[c]} | {busy:.3f} | {lanes:.1f} | {busy * lanes / 64:.2f} | {d['wave_cycles_share']['wait_any']:.2f} | "
          f"{d['wave_instructions_per_path']['vmem_rd']:.2f} | {d['hbm_bytes_per_path']:.1f} | {d['wave_instructions_per_path']['valu']:.0f} |")
print()
print("| config | frame in the bench line | Msamples/s | segments per sample | CPU Msamples/s | ratio |")
print("|---|---|---|---|---|---|")
for c, f, frame in (("C2", f"{tag}_bench.json", "1200x800"), ("C3", f"{tag}_bench_C3.json", "1920x1080"), ("C4", f"{tag}_bench_C4.json", "1024x1024"),
                    ("C5", f"{tag}_bench_C5.json", "3840x2160")):
    d = load(f)
    cpu = d["cpu_baseline"]["value"]
    print(f"| {names[c]} {frame} | {d['config']['spp']} spp | {d['value']:.0f} | {d['segments_per_sample']:.2f} | {cpu:.2f} | {d['value'] / cpu:.0f} |")
d = load(f"{tag}_bench.json")
r = d["roofline"]
print()
print("C2 line: value %.0f, ms_per_step %.2f, avg_launch_ms %.2f, roofline.frac %.3f (busy %.3f x lanes %.1f / 64), pipeline_equivalent %.0f GB/s = %.3f, traffic %.3g B = hbm_frac %.3f, valu_f64 %.1f Top/s = %.2f, build %s, stale %s" %
      (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"] or -1, r.get("valu_issue_busy") or -1, r.get("lanes_active_per_valu_instruction") or -1,
       r["pipeline_equivalent"]["achieved"], r["pipeline_equivalent"]["frac"], r["traffic"] or -1, r.get("hbm_frac_measured") or -1,
       r["valu_f64"]["achieved_tops"], r["valu_f64"]["frac"], d["library_build_id"], r.get("counters_stale")))
print("cpu baseline:", json.dumps(d["cpu_baseline"]))
print("kernel_ms:", d["kernel_ms"], "scene_upload_ms %.2f first_call_ms %.0f" % (d["scene_upload_ms"], d["library_first_call_ms"]))
for f in (f"{tag}_bench_wavefront.json", f"{tag}_bench_paths_64spp.json"):
    w = load(f)
    print(f, round(w["value"], 1), w.get("kernel_ms"))
