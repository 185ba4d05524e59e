#!/bin/bash
# developer build only (DBFR_BUILD_DEV=1): what each part of k_convz costs (results wrong except ABL=0)
for a in 0 1 2 4 8 16 32 12 5; do
  echo "== DBFR_CONVZ_ABL=$a"
  DBFR_CONVZ_ABL=$a python tools/exp/convz_check.py --timeonly 2>&1 | grep reduce_first
done
