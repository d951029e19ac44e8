"""What bench.py measures WITH: the synthetic workloads (generated and packed on the device, untimed setup), the launcher plumbing,
the N > 1 drivers that touch no oracle (`--exchange c`: one process for all devices; `--emulate-world`: one GPU as one rank of N), and
-- since round 6 -- the legs beside the headline (benchlib/legs.py: the CPU baseline with the end-to-end commands, the device tokeniser's
leg, the fastq and mammal workloads with their parity checks).  bench.py keeps the headline's timed loop, its roofline arithmetic and
the line."""
