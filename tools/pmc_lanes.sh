#!/bin/bash
# Lanes at work per vector instruction, per kernel (one rocprofv3 --pmc run): SQ_THREAD_CYCLES_VALU over
# SQ_ACTIVE_INST_VALU, to be read against a kernel whose waves are full (resolveKernel, generateKernel).
#   tools/pmc_lanes.sh WORKLOAD [bench args...] -> gpurun_out/pmc_lanes_<workload>.txt
WL=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmclanes_$WL
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE \
    --output-format csv -d gpurun_out/pmclanes_$WL -o p -- python3 bench.py --workload $WL --steps 1 --warmup 0 --pmc-pass "$@" > gpurun_out/pmc_lanes_$WL.log 2>&1 || { tail -5 gpurun_out/pmc_lanes_$WL.log; exit 1; }
python3 tools/pmc_summary.py gpurun_out/pmclanes_$WL/p_counter_collection.csv > gpurun_out/pmc_lanes_$WL.txt
rm -rf gpurun_out/pmclanes_$WL
python3 - gpurun_out/pmc_lanes_$WL.txt <<'PY'
import re, sys
for line in open(sys.argv[1]):
    m = re.match(r"(\S+(?:<[^>]*>)?)\s+n=", line)
    a = re.search(r"SQ_ACTIVE_INST_VALU/disp=(\S+)", line)
    t = re.search(r"SQ_THREAD_CYCLES_VALU/disp=(\S+)", line)
    i = re.search(r"SQ_INSTS_VALU/disp=(\S+)", line)
    if m and a and t and i:
        print("%-44s thread-cycles / active-inst = %6.2f   thread-cycles / instruction = %6.2f" % (line.split(" n=")[0].strip(), float(t.group(1)) / float(a.group(1)), float(t.group(1)) / float(i.group(1))))
PY
