"""`paddle` compat namespace (SURVEY.md §8(b) adapter (ii), Appendix A): the Paddle symbols the reference's
models/rank nets, dygraph_model.py files and tools/trainer.py touch, with the embedding lookup, the Linear GEMMs, the
SelectedRows merge and the Adam / SGD updates served by the recengine HIP kernels through the C-ABI.

    python -m paddlerec_amd.run_reference /path/to/PaddleRec/tools/trainer.py -m models/rank/deepfm/config.yaml

puts this directory in front of sys.path, so that the reference's UNMODIFIED scripts `import paddle` from here.
Tensors are torch tensors (device memory and the autograd tape: plumbing); the operators with a kernel in
librecengine.so go through paddlerec_amd.ops — with the HIP library missing or tensors on the CPU they fail loudly
unless an operator backend is named in REC_COMPAT_KERNELS (tests run the host logic with an oracle-backed stand-in)."""
