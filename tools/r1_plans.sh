#!/bin/bash
# Developer sweep of the one-row kernel's class-1 source plans: objects prebuilt under tools/ubench/_bin/plans/ (CHANGELOG, round 4 notes),
# relinked into the library one at a time on the GPU box and timed with tools/quick_bench.py.
cd $GRAFT_REPO_ROOT/soc-matching_amd/csrc
for o in ../../tools/ubench/_bin/plans/r1_*.o; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../socmx/libsocmx.so socmx_baselines.o socmx_loss.o socmx_rollout.o $o socmx_rollout_ctrl.o socmx_stopping.o socmx_unet_bwd.o || exit 1
  echo "== $(basename $o .o)"
  (cd ../.. && python3 tools/quick_bench.py cfg3 cfg2 2>&1 | grep -E "parity|rollout")
done
