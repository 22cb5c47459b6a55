#!/bin/bash
# PMC counters of the fast traversal kernel on the ray-set micro-benchmark.
# usage: tools/pmc_trav.sh <outdir> [config]
OUT=$1; CFG=${2:-C2}
export TMPDIR=/tmp
mkdir -p $OUT
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python3 tools/trav_bench.py $CFG --reps 2 > $OUT/$name.log 2>&1 || exit 1
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS
run b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run c SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY
python3 - <<PY
import csv, glob, collections
for path in sorted(glob.glob("$OUT/*/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(path)):
        if "k_traverse" in row["Kernel_Name"] and "IntersectIo" in row["Kernel_Name"]:
            a = agg[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    for k, (v, c) in sorted(agg.items()):
        print(f"{k:28s} {v / c:16.0f}  (per dispatch, {c} dispatches)")
PY
