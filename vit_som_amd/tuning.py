"""Test and measurement hooks -- NOT a configuration surface.

The product always runs with these defaults; the GPU suite flips them to prove that the stream-level
concurrency, the forward split and the overlapped exchange leave the result bit-identical, and the A/B tools
under tools/ use them to time one variant against another inside one process.  (Round 1 read environment
variables at these points, some of them inside the C library on every launch.)"""


class _Hooks:
    side_stream = True          # weight-gradient GEMMs and the SOM backward on side streams
    fwd_split = True            # forward as two half-batch chains on two streams
    fwd_split_blocks = None     # how many encoder blocks run as two chains (None = all)
    overlap_allreduce = True    # N > 1: issue the all-reduce pieces inside the backward pass
    bucket_blocks = 3           # encoder blocks per early all-reduce piece
    bmu_planes = True           # cosine BMU pass on pre-split plane images where the shape allows (model.py SOMLayer._distances_into)
    adamw_planes = False        # True: FusedAdamW writes the prototypes' plane image in its own pass (+27 us on the critical path);
                                # False: the training forward re-splits them on the SOM stream under the encoder (hidden)
    ln_reduce_batched = True    # the LayerNorm backwards' dgamma / dbeta reductions in one launch per exchange piece instead of 30 (ops.LayerNormJobs)
    bmu_overlap = False         # True: the BMU pass on the SOM stream under the decoder forward (model.py _run_forward): -0.02..-0.04 ms per
                                # step, but the contraction then shares the chip (72 instead of 58-63 us per launch) -- off: it runs alone
    launch_tape = True          # train_step_fused / training_step re-issue the recorded launches of a step from C (model.py)

    def set(self, **kw):
        for k, v in kw.items():
            if not hasattr(_Hooks, k):
                raise AttributeError(f"unknown hook {k!r}")
            setattr(self, k, v)
        return self

    def signature(self):
        """All switches as a tuple: what a recorded launch sequence was recorded under (model.py _tape_key)."""
        return tuple((k, getattr(self, k)) for k in sorted(vars(_Hooks)) if not k.startswith("_") and not callable(getattr(_Hooks, k)))

    def reset(self):
        for k in list(self.__dict__):
            delattr(self, k)
        return self


hooks = _Hooks()
