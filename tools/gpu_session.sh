#!/bin/bash
TAG=${TAG:-r02f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
run_pmc() {  # $1 = label, $2.. = env assignments ; counters in $C
  label=$1; shift
  i=0
  for c in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_BUBBLE_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    (cd /tmp && env "$@" timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_${label}_$i -- python "$GRAFT_REPO_ROOT/tools/pmc_grid_fwd.py" > /tmp/pmc_${label}_$i.log 2>&1)
    f=$(find /tmp/pmc_${label}_$i -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python tools/pmc_summary.py $f | grep -i "grid_fwd" | sed 's/.*",//; s/^_ZN[^,]*,//' >> $OUT/pmc_$label.csv; else tail -3 /tmp/pmc_${label}_$i.log >> $OUT/pmc_$label.err; fi
  done
  # kernel duration from the trace of the last pass
  k=$(find /tmp/pmc_${label}_$i -name "*kernel_trace.csv" | head -1)
  [ -n "$k" ] && python - "$k" >> $OUT/pmc_$label.csv <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "grid_fwd" in r["Kernel_Name"]]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows][5:]
print("duration_us_mean,%d,%.2f" % (len(d), sum(d)/len(d)))
PY
  echo "== $label"; cat $OUT/pmc_$label.csv
}
run_pmc plain PVD_GRID_LPS=0
run_pmc lps2 PVD_GRID_LPS=2
run_pmc lps2p4k PVD_GRID_LPS=2 PVD_GRID_PERSIST=4096
