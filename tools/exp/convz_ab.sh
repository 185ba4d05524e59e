#!/bin/bash
# developer: convz_check --timeonly through every variant library tools/exp/ab/libdbfr_<name>.so named on the command line, and the tree's ("tree")
R=$GRAFT_REPO_ROOT; cd $R
for v in "$@"; do
  if [ $v = tree ]; then unset DBFR_LIB; else export DBFR_LIB=$R/tools/exp/ab/libdbfr_$v.so; fi
  for r in 1 2; do echo "== $v r$r"; timeout 200 python tools/exp/convz_check.py --timeonly 2>&1 | grep reduce_first; done
done
