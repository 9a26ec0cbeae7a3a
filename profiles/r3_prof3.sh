#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
timeout 180 python profiles/mega_prof.py 4 2>&1 | grep -v amdgpu.ids | grep -E "trace_ms|span|slot time|pcts|last waves|wave [0-9]+ tile|in flight|>= 256" | head -40
