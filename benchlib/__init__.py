"""What bench.py measures WITH, none of it measurement: the synthetic workloads (generated and packed on the device, untimed
setup) and the launcher plumbing.  The timed loops, the roofline arithmetic and every use of the oracle / the reference binary
(the CPU baseline and the parity checks of the legs) stay in bench.py."""
