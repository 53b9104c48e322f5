#!/usr/bin/env bash
# A/B of two builds on ONE GPU box: the tree itself against a copy of another commit in ./ab_old (git worktree add -f ab_old <commit>; build it
# there): runs "$@" alternately in both, three times.   tools/ab_run.sh python tools/klt_layout_ab.py 64
for i in 1 2 3; do
  echo "== old"; ( cd ab_old && "$@" )
  echo "== new"; "$@"
done
