mkdir -p gpurun_out/r6
for rep in 1 2; do for v in "" "_nores2"; do echo "== variant '$v' rep $rep"; if [ -n "$v" ]; then export SOCMX_LIB=soc-matching_amd/socmx/libsocmx$v.so; else unset SOCMX_LIB; fi; python3 tools/quick_bench.py cfg3 cfg2 oul10 ouq20 md 2>&1 | grep -E "parity|rollout"; done; done
