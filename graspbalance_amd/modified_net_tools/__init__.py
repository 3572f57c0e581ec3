"""Counterparts of the reference's ModifiedNetTools/ (OpenPoints-style layer factories and the
PB-ext wrappers group.py / subsample.py / upsampling.py) over the HIP-backed ``pointnet2_batch_cuda``."""
