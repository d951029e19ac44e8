#!/bin/bash
# development helper (GPU box): scan kernel variants side by side on the same random batch
cd $GRAFT_REPO_ROOT
for m in ${MODES:-0 4 5}; do
 for a in ${ABL:-0}; do
  echo -n "ablate=$a "; KSSD_DEV_ABLATE=$a KSSD_DEV_SCAN=$m timeout 120 profiles/scanbench ${SB_ARGS:-400 5000000 10}
 done
done
