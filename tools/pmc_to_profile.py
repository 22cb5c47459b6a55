"""Turns tools/pmc_summary.py output into the per-launch record kept under profiles/.
usage: python tools/pmc_to_profile.py <pmc_summary.json> <paths_per_launch> <command string> > profiles/rNN_pmc_k_paths.json"""
import json
import sys


def main(path, paths_per_launch, command):
    with open(path) as f:
        summary = json.load(f)
    name = next(k for k in summary if "k_paths" in k)
    c = summary[name]
    n = c.pop("_dispatches")
    per_launch = {k: v / n for k, v in c.items()}
    # FETCH_SIZE / WRITE_SIZE come in KiB; on gfx950 a 128-byte read request counts as 64 bytes
    hbm = (2.0 * per_launch["FETCH_SIZE"] + per_launch["WRITE_SIZE"]) * 1024.0
    import os

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from ray_tracer_amd import _abi

    out = {
        # the library these counters were collected with (this script runs in the same tree, right after the passes)
        "library_build_id": _abi.libmirt().mirt_build_id().decode(),
        "command": command,
        "kernel": name,
        "paths_per_launch": paths_per_launch,
        "counters_per_launch": dict(sorted(per_launch.items()), _dispatches=n),
        "units": "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; SQ_*_CYCLES and SQ_ACTIVE_INST_* in quad-cycles",
        "hbm_bytes_per_launch": hbm,
        "hbm_bytes_note": "FETCH_SIZE doubled (gfx950 counts 128-byte read requests as 64 bytes); WRITE_SIZE as reported",
        "hbm_bytes_per_path": hbm / paths_per_launch,
        "lanes_active_per_valu_instruction": per_launch["SQ_THREAD_CYCLES_VALU"] / per_launch["SQ_INSTS_VALU"],
        # every wave lives as long as the kernel, SQ_WAVES / 1024 of them share one of the 1024 SIMDs: the
        # SIMD's cycles are the wave cycles divided by that, and a SIMD issues one vector instruction at a time
        "valu_issue_busy_share_of_simd_cycles": per_launch["SQ_ACTIVE_INST_VALU"] / (per_launch["SQ_WAVE_CYCLES"] / (per_launch["SQ_WAVES"] / 1024.0)),
        "valu_instructions_per_path": per_launch["SQ_INSTS_VALU"] * 64.0 / paths_per_launch,
        "wave_instructions_per_path": {k: per_launch[c] / paths_per_launch for k, c in
                                       (("valu", "SQ_INSTS_VALU"), ("salu", "SQ_INSTS_SALU"), ("lds", "SQ_INSTS_LDS"), ("vmem_rd", "SQ_INSTS_VMEM_RD"))
                                       if c in per_launch},
        "wave_cycles_share": {k: per_launch[c] / per_launch["SQ_WAVE_CYCLES"] for k, c in
                              (("wait_any", "SQ_WAIT_ANY"), ("wait_inst_any", "SQ_WAIT_INST_ANY"), ("active_inst_any", "SQ_ACTIVE_INST_ANY"))
                              if c in per_launch},
        "lds_bank_conflict_share_of_lds_active": (per_launch["SQ_LDS_BANK_CONFLICT"] / per_launch["SQ_ACTIVE_INST_LDS"]
                                                  if per_launch.get("SQ_ACTIVE_INST_LDS") else None),
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3])
