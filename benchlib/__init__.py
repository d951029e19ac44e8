"""What bench.py measures WITH: the synthetic workloads (generated and packed on the device, untimed setup), the launcher plumbing,
and the N > 1 drivers that touch no oracle (`--exchange c`: one process for all devices; `--emulate-world`: one GPU as one rank of N).
The headline's timed loop, the roofline arithmetic and every use of the oracle / the reference binary (the CPU baseline and the
parity checks of the legs) stay in bench.py."""
