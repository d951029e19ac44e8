#!/bin/bash
# the -A case of profiles/r06long_fuzz_cli.txt (case 30, seed 1210030) through both binaries
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
D=$(mktemp -d /dev/shm/koc_XXXX)
cd "$D" && mkdir in && cp "$R/profiles/cases/r06_koc/f00.fastq.gz" in/
"$R/public_kssd_amd/kssd" shuffle -k 9 -s 6 -l 3 -o p --seed 1030 > /dev/null
for p in 1 2 4 16; do
  for rep in 1 2 3; do
    rm -rf o_ref; "$R/oracle/_ref/kssd" dist -p $p -L p.shuf -A -o o_ref in > /dev/null 2>&1
    echo "ref p$p: $(od -A n -t u4 o_ref/combco.0 | tr -s ' \n' ' ') | $(od -A n -t u2 o_ref/combco.0.a | tr -s ' \n' ' ')"
  done
  rm -rf o_our; "$R/public_kssd_amd/kssd" dist -p $p -L p.shuf -A -o o_our in > our.log 2>&1; echo "rc $?"
  echo "our p$p: $(od -A n -t u4 o_our/combco.0 | tr -s ' \n' ' ') | $(od -A n -t u2 o_our/combco.0.a | tr -s ' \n' ' ')"
done
tail -5 our.log
KSSD_HOST_FASTQ=1 "$R/public_kssd_amd/kssd" dist -p 2 -L p.shuf -A -o o_host in > /dev/null 2>&1
echo "our host-tokenised: $(od -A n -t u4 o_host/combco.0 | tr -s ' \n' ' ') | $(od -A n -t u2 o_host/combco.0.a | tr -s ' \n' ' ')"
cd /; rm -rf "$D"
