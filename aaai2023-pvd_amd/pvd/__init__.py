"""Host harness around the HIP hot path: the counterparts of the reference's
``NeRFRenderer.run_cuda`` / ``NeRFNetwork`` / ``Trainer.train_step`` that the metric needs.
Not a re-implementation of PVD as a product (trainer, data provider and CLI
are out of scope); it exists so that bench.py can run the distillation / teacher step and the
parity tests can exercise the operators in their real call pattern."""
