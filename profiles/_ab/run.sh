#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for rep in 1 2; do for v in A B; do
  echo "== $v (run $rep)"; YCGE_LIB=$REPO/profiles/_ab/lib$v.so python profiles/post_nosky.py 2>&1 | tail -2
  YCGE_LIB=$REPO/profiles/_ab/lib$v.so python bench.py --steps 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg4', d['post_stage']['post_ms'])"
done; done
