#!/bin/bash
# round-2 GPU call 1: GEMM tile A/B (standalone harness), effective clock / MFMA-busy counters, full-width parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2a; mkdir -p $O
B=tools/probe/gemm_bench
timeout 300 $B --rounds 9 \
  g:4096,4096,4096,24 g:4096,4096,4096,26 g:4096,4096,4096,22 g:4096,4096,4096,0 g:4096,4096,4096,9 \
  g:8192,8192,4096,24 g:8192,8192,4096,26 \
  g:767,12288,4096,24 g:767,12288,4096,26 g:767,21760,4096,24 g:767,21760,4096,26 \
  g:767,4096,11008,24,4 g:767,4096,11008,26,4 g:767,4096,11008,26,5 g:767,4096,11008,7 \
  g:767,4096,4096,7 g:767,4096,4096,0 g:767,4096,4096,24,2 g:767,4096,4096,26,2 g:767,4096,4096,26,4 \
  c:1,192,192,1024,24 c:1,192,192,1024,26 c:1,96,96,1024,24 c:1,96,96,1024,26 c:1,48,48,1024,4,2 c:1,48,48,1024,12,2 \
  > $O/gemm_big_u.jsonl 2> $O/gemm_big_u.err
timeout 120 $B --rounds 9 --fill z g:4096,4096,4096,24 g:4096,4096,4096,26 g:8192,8192,4096,24 g:8192,8192,4096,26 \
  > $O/gemm_big_z.jsonl 2>> $O/gemm_big_u.err
timeout 300 $B --rounds 15 \
  g:577,3072,1024,4 g:577,3072,1024,12 g:577,3072,1024,13 g:577,3072,1024,14 g:577,3072,1024,15 g:577,3072,1024,0 g:577,3072,1024,11 \
  g:577,4096,1024,4 g:577,4096,1024,12 g:577,4096,1024,13 g:577,4096,1024,14 g:577,4096,1024,15 g:577,4096,1024,0 \
  g:577,1024,1024,4 g:577,1024,1024,12 g:577,1024,1024,14 g:577,1024,1024,14,2 g:577,1024,1024,13,2 \
  g:577,1024,4096,4,3 g:577,1024,4096,12 g:577,1024,4096,12,2 g:577,1024,4096,14 g:577,1024,4096,14,2 g:577,1024,4096,13,2 \
  g:4616,3072,1024,0 g:4616,3072,1024,24 g:4616,3072,1024,26 g:4616,4096,1024,24 g:4616,4096,1024,26 g:4616,1024,4096,0 g:4616,1024,4096,26 g:4616,1024,1024,0 \
  > $O/gemm_vit_u.jsonl 2> $O/gemm_vit_u.err
# effective clock and MFMA busy of the two 256x256 kernels (counters in their own pass, kernel-trace only)
( cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_mfma --output-format csv -- \
    $GRAFT_REPO_ROOT/$B --rounds 3 g:4096,4096,4096,24 g:4096,4096,4096,26 g:767,12288,4096,24 g:767,12288,4096,26 c:1,192,192,1024,24 c:1,192,192,1024,26 ) > $O/pmc_mfma.log 2>&1
( cd /tmp && rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_sq --output-format csv -- \
    $GRAFT_REPO_ROOT/$B --rounds 3 g:4096,4096,4096,24 g:4096,4096,4096,26 ) > $O/pmc_sq.log 2>&1
timeout 1500 python -m pytest tests/test_fullwidth_gpu.py tests/test_kernels_gpu.py -q -m gpu -x -k "fullwidth or production or llama_7b or config1 or gemm or conv3x3" -s > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
cat $O/gemm_big_u.jsonl | cut -c1-200
