#!/bin/bash
# HISTORICAL (round 5): archived with the experiment it drove.  It expects the layout of that round -- this script, r1p_check.py
# and r1p_mkplan.py under tools/, socmx_rollout1p.hip under soc-matching_amd/csrc/ -- and does not run from here as it stands:
# to reproduce, copy socmx_rollout1p*.hip into soc-matching_amd/csrc/ and these scripts into tools/ first (README.md, "Reproducing").
# Developer sweep of the packed-fma one-row kernel's source plans (csrc/socmx_rollout1p.hip: SOCMX_R1P_PLAN0 / 1): each plan is
# compiled into ITS OWN library under tools/ubench/_bin/plans/ -- only socmx_rollout1p.o differs, the other objects are the shipped
# build's; the shipped soc-matching_amd/socmx/libsocmx.so is never touched -- and loaded through SOCMX_LIB (socmx/_lib.py).
#   prebuild (no GPU needed):  bash tools/r1_plans.sh build NAME PLAN0 PLAN1 [extra compiler flags]      (plans: tools/r1p_mkplan.py)
#   on the GPU box:            bash tools/r1_plans.sh run
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
P=$ROOT/tools/ubench/_bin/plans
C=$ROOT/soc-matching_amd/csrc
mkdir -p $P
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -c $C/socmx_rollout1p.hip -o $P/r1p_$2.o \
     "-DSOCMX_R1P_PLAN0=\"$3\"" "-DSOCMX_R1P_PLAN1=\"$4\"" $5 -Rpass-analysis=kernel-resource-usage 2>&1 \
     | grep -E "error|VGPRs Spill" | sed 's/.*VGPRs Spill: //; s/\[-R.*//' | tr '\n' ' '
  echo " <- VGPR spills per instantiation ($2)"
  OBJS=$(ls $C/*.o | grep -v socmx_rollout1p.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libsocmx_$2.so $OBJS $P/r1p_$2.o && rm -f $P/r1p_$2.o
else
  for l in $P/libsocmx_*.so; do
    echo "== $(basename $l .so): $(cd $ROOT && SOCMX_LIB=$l python3 tools/r1p_check.py time-only 2>/dev/null | tail -1) ms"
  done
fi
