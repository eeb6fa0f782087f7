#!/usr/bin/env python3
"""Index arithmetic of the quad-cooperative node fetch (csrc/pt_scene.h CoopFetchNodesQ / TravNodeStepQCoop, k_gather_probe_coop<2>)
modelled for one 64-lane wave: ds_bpermute of the record index, LDS-DMA placement (wave-uniform base + lane x 16), the producer-side word rotation
and the 4 x ds_read_b128.  Checks that every lane ends up with the four words of ITS record and that the 16 lanes an LDS
pass serves touch 16 distinct 16-byte bank groups for every read instruction.  (A model of the code, not the code: the GPU-side check is
mi_gather_rate_coop's lanes_equal.)"""
import numpy as np

rng = np.random.default_rng(1)
nrec = 1000
buf = rng.integers(0, 2**32, size=(nrec, 4, 4), dtype=np.uint64)   # record, 16-byte word, dword
rec = rng.integers(0, nrec, size=64)                               # each lane's own record
lane = np.arange(64)
stage = np.zeros((4, 64, 4), dtype=np.uint64)                      # uint4 stage[4][64]
for k in range(4):                                                 # instruction k serves the records of lanes 16k .. 16k+15
    owner = (lane >> 2) + 16 * k
    r = rec[owner]                                                 # ds_bpermute(owner * 4, rec)
    stage[k, lane, :] = buf[r, ((lane & 3) + (lane >> 4)) & 3, :]   # global_load_lds_dwordx4: &stage[k][0] + lane * 16; producers rotate the word order
flat = stage.reshape(256, 4)
for L in range(64):
    base, rot = 64 * (L >> 4) + 4 * (L & 15), (L >> 2) & 3
    w = [flat[base + ((i - rot) & 3)] for i in range(4)]           # word i sits at position (i - rot) & 3
    assert all((w[j] == buf[rec[L], j]).all() for j in range(4)), L
for j in range(4):
    for g in range(4):
        groups = {((64 * (L >> 4) + 4 * (L & 15) + ((j - ((L >> 2) & 3)) & 3)) % 16) for L in range(16 * g, 16 * g + 16)}
        assert len(groups) == 16, (j, g)
print("exchange exact for all 64 lanes; every ds_read_b128 pass of 16 lanes touches 16 distinct 16-byte bank groups")
