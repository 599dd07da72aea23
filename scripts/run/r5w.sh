#!/bin/bash
# SQ counters of the MSD transforms with twiddles from registers (this tree) and from LDS tables (the header of the
# commit before 6a848c7, shipped as scripts/diag/_table_twiddles_msd_fft.hpp.txt): LDS / VALU instructions per launch
out=$GRAFT_REPO_ROOT/gpurun_out/r5w; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
pass() { tag=$1; shift
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $out/${tag}_insts -o $tag -- python3 $GRAFT_REPO_ROOT/bench.py --workload msd --steps 2 --warmup 1 --no-onsager --no-cpu-baseline "$@" > $out/${tag}_insts.json 2> $out/${tag}_insts.err; echo "$tag insts rc=$?"
  timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/${tag}_lds -o $tag -- python3 $GRAFT_REPO_ROOT/bench.py --workload msd --steps 2 --warmup 1 --no-onsager --no-cpu-baseline "$@" > $out/${tag}_lds.json 2> $out/${tag}_lds.err; echo "$tag lds rc=$?"
}
pass new_b1
pass new_b250 --blocks 250
cd $GRAFT_REPO_ROOT
cp mdhelper_amd/csrc/mdx_msd_fft.hpp $out/new.hpp
cp scripts/diag/_table_twiddles_msd_fft.hpp.txt mdhelper_amd/csrc/mdx_msd_fft.hpp
make -C mdhelper_amd/csrc > $out/make_old.log 2>&1; echo "make rc=$?"
cd /tmp
pass old_b1
pass old_b250 --blocks 250
cd $GRAFT_REPO_ROOT
cp $out/new.hpp mdhelper_amd/csrc/mdx_msd_fft.hpp; rm $out/new.hpp
python - <<'PY'
import csv, glob, collections
for tag in ("old_b1", "new_b1", "old_b250", "new_b250"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
    for kind in ("insts", "lds"):
        for path in glob.glob(f"gpurun_out/r5w/{tag}_{kind}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(path)):
                k = row["Kernel_Name"]
                if "msd_fft" not in k: continue
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); calls[(k, kind)].add(row["Dispatch_Id"])
    for k, c in agg.items():
        n = max(len(calls[(k, "insts")]), 1)
        print(tag, k.split("(")[0][-46:], "launches", n, {a: "%.4g" % (b / n) for a, b in sorted(c.items())})
PY
