"""Host-side mirror of the reference's `gnt` package for the attack path (GNT flavour, config 4): the per-ray network and
renderer run on the HIP kernels of csrc/nf_gnt.hip; projection, ray sampling and the ResUNet are shared with the IBRNet
flavour (the reference's gnt/projection.py, sample_ray.py, feature_network.py are copies with the differences noted there)."""
