#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2e; mkdir -p $O
B=tools/probe/gemm_bench
timeout 300 $B --rounds 9 g:4096,4096,4096,24 g:4096,4096,4096,26 g:4096,4096,4096,22 g:4096,4096,4096,0 \
  g:767,12288,4096,24 g:767,12288,4096,26 g:767,21760,4096,24 g:767,21760,4096,26 g:767,21760,4096,24,1,0,1 \
  g:767,4096,11008,24,4 g:767,4096,11008,26,5 g:767,4096,11008,26,4 g:767,4096,4096,7 g:767,4096,4096,26,4 g:767,4096,4096,24,3 \
  g:4616,3072,1024,24 g:4616,3072,1024,0 c:1,192,192,1024,24 c:1,192,192,1024,26 c:1,96,96,1024,24 > $O/gemm_epi.jsonl 2> $O/gemm_epi.err
cut -c1-210 $O/gemm_epi.jsonl
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_fullwidth_gpu.py -q -m gpu -k "gemm or conv3x3 or production or config3 or config5" > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
grep -E "passed|failed|rc |^FAILED|^E  " $O/pytest.log | tail -20 | cut -c1-220
