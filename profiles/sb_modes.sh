#!/bin/bash
# development helper (GPU box): the scan kernel with parts compiled out, on the same random batch
#   ABL="1 2 3 0" profiles/sb_modes.sh      1 = loads only, 2 = + stage 1, 3 = + candidate capture and Bloom rounds
#                                           without the candidate list, 0 = the product kernel
cd $GRAFT_REPO_ROOT
for a in ${ABL:-1 2 3 0}; do
  echo -n "ablate=$a "; KSSD_DEV_ABLATE=$a timeout 120 profiles/scanbench ${SB_ARGS:-400 5000000 10} | grep variant
done
