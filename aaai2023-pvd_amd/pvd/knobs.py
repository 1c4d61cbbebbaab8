"""Registry of the PVD_* environment switches: every name product code reads is listed here with its default, what it selects and
who sets it (tests/test_host_logic.py::test_every_environment_switch_is_registered greps the sources against this table).

Two kinds.  PRODUCTION switches a deployment may set.  TEST switches force an alternative code path that the test-suite proves
equivalent to the default (or a failure the fall-backs must survive); they carry no performance meaning and nothing in bench.py's
default run sets them.  Measurement-only knobs whose A/B is closed are gone (their records are under profiles/; see
profiles/HISTORY.md sections 5 and 11.7 for the lists removed in rounds 5 and 6)."""

PRODUCTION = {
    "PVD_HIP_LIB": ("<package>/libpvd_hip.so", "path of an A/B build of the same C ABI (pvd_hip/__init__.py)"),
    "PVD_DIST_BACKEND": ("nccl", "nccl (= RCCL, one GPU per rank) | gloo (ranks may share a GPU: how the GPU tests run two ranks on a one-GPU box)"),
    "PVD_DP_EXCHANGE": ("allreduce", "ray-DP gradient exchange: allreduce (gather that zeroes / checks -> all-reduce -> the update reads the buffer) | "
                                     "sharded (reduce_scatter -> AdamW on the rank's rows -> all_gather) | classic (rounds 1-5's sequence, incl. the objective's "
                                     "four separate launches): pvd/trainer.py, pvd/ray_dp.py"),
    "PVD_DP_HASH_WIRE": ("f16", "f16 | f32: a hash student's table gradient under ray-DP crosses the links as the half-precision table the scatter wrote "
                                "(the reference's arithmetic for this gradient; 21 MB) or widened into the fp32 bucket first (42 MB; rounds 1-5)"),
    "PVD_DP_WIRE": ("f32", "f32 | f16 | bf16: width of the gradient on the wire (classic form only; measured -0.3 dB PSNR: opt-in)"),
    "PVD_STEPS_PER_GRAPH": ("", "bench.py: steps recorded per hipGraph launch (default: the largest of 20/10/5/4/2 dividing --steps)"),
    "PVD_HW_QUEUES": ("", "keep = do not set GPU_MAX_HW_QUEUES=2 at import (an integrator who owns that variable); the forked schedule is then refused "
                          "unless the caller exported 2 itself"),
}

TEST = {
    "PVD_DP_FORCE": ("0", "1 = run the ray-DP code path (collectives, exchange forms) in a world of ONE rank: RCCL next to graph capture on a one-GPU box"),
    "PVD_DP_PIPELINE": ("1", "0 | 1 (fork the next step's prefix when there is more than one rank) | 2 (always: the one-rank tests)"),
    "PVD_DP_INGRAPH": ("1", "0 = keep RCCL's collectives out of the graphs (graphs cut at the collectives: the gloo form, and the fall-back of a failed capture)"),
    "PVD_DP_OVERLAP": ("1", "segmented form only: 0 = do not replay the next step's prefix under the eager exchange"),
    "PVD_DP_SHARDED_MIN": ("65536", "elements below which a sharded sum stays a plain all-reduce (tests lower it for toy models)"),
    "PVD_ADAMW_SPLIT": ("late", "late | 0: AdamW in two parts (L1-only rows one step later on the branch) or one launch -- bit-identical (tests/test_hip_graph.py)"),
    "PVD_ADAMW_LAZY": ("1", "0 = decay the cold parameter groups every step instead of logging and replaying the decays -- bit-identical"),
    "PVD_TOUCHED_SET": ("1", "0 = zero / check the whole gradient buffer instead of the rows the occupancy grid lets a sample touch -- same training"),
    "PVD_INF_CHECK_RIDE": ("1", "0 = GradScaler's inf check as a launch of its own instead of riding on the table scatter / the exchange"),
    "PVD_HEAD_DW_RIDE": ("1", "0 = the VM head's riders on the lookup's launches as launches of their own: the weight-gradient reduction (extra workgroups of the table scatter) and the f16 weight image (extra workgroups of the lookup's forward)"),
    "PVD_MLP_FUSED": ("1", "0 = the frozen NeRF-MLP teacher layer by layer instead of pvd_mlp_head_forward_fused"),
    "PVD_INFER_PERSISTENT": ("1", "0 = inference as rounds (march / forward / composite / compact launches) instead of one persistent launch -- bit-identical images"),
    "PVD_INFER_DEVICE_ROUNDS": ("1", "0 = the reference-shaped host loop with a read-back per round"),
    "PVD_INFER_SHUFFLE": ("", "C side: slot-assignment stride of the persistent renders (a permutation of the work order; images bit-identical)"),
    "PVD_FUSED_DMA": ("", "C side: 0 / 1 = the fused lookup's weight image by register copies / LDS-DMA -- bit-identical"),
    "PVD_FORKED_GRAPHS": ("", "0 | 1 overrides whether two-chain hipGraphs may be recorded (default: only with GPU_MAX_HW_QUEUES=2)"),
    "PVD_CAPTURE_GC": ("1", "0 = leave the cyclic garbage collector on while recording (the round-2 SIGABRT's regression test)"),
    "PVD_TEST_FAIL_IN_CAPTURE": ("", "1 | forked = raise inside a recording: exercises every fall-back"),
}

ALL = {**PRODUCTION, **TEST}
