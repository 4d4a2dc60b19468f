#!/bin/bash
# usage (GPU box): tools/gpu_poison.sh [pytest arguments]  -- the GPU tests with every device allocation of the library filled with 0xFF bytes
# (SEERHIP_DEBUG=poison, csrc/api.hip dmalloc): a kernel that reads memory nobody wrote sees NaN / -1 instead of whatever the allocator left
# there (a fresh process gets zeroed pages, a long test session does not).
cd "$GRAFT_REPO_ROOT"
export SEERHIP_DEBUG=poison
timeout 1700 python -m pytest ${@:-tests/test_glm_gpu.py tests/test_lmm_gpu.py tests/test_job_gpu.py} -q -p no:cacheprovider 2>&1 | tail -40
