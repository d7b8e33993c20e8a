#!/bin/bash
# GPU box: SQ stall counters of pod_wino_conv3x3 (bench launch) for a list of library builds: tools/wino_pmc2.sh "<tag> <tag> ..."  ("." = shipped)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc2
for tag in $1; do
  lib=pod_compare_amd/lib/$tag/libpod_mi355x.so; [ "$tag" = "." ] && lib=pod_compare_amd/lib/libpod_mi355x.so
  for counters in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_IFETCH SQ_WAIT_IFETCH" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_sum"; do
    raw=gpurun_out/pmc2/raw_pmc; rm -rf $raw
    POD_MI355X_LIB=$lib timeout 300 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $raw -o p -- python tools/wino_only.py 2 19 bench > /dev/null 2>&1
    f=$(find $raw -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python - "$f" "$tag" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "wino_conv3x3" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print("%s %s %.6g" % (sys.argv[2], k, sum(v) / len(v)))
PY
    rm -rf $raw
  done
done
