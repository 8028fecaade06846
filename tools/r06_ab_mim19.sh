#!/bin/bash
# interleaved A/B of the mim_19 step: product library vs the one with the round-5 table entry for qkv fwd
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then export SKYEMB_LIB=$PWD/sky_embeddings_amd/libskyemb_oldtab.so; else unset SKYEMB_LIB; fi
    SKYEMB_BENCH_NO_AB=1 timeout -k 10 200 python tools/mim19_bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],3))"
  done
done
