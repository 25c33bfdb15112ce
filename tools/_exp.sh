#!/bin/bash
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -o /tmp/cb tools/ubench/contraction_bench.hip 2>/dev/null
/tmp/cb 64 400 512 5; /tmp/cb 32 100 512 5; /tmp/cb 20 50 256 5
python -m pytest tests/test_gpu_parity.py -q -x -k "target or contraction or full_size or fused or golden" 2>&1 | tail -3
