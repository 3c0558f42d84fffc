"""Test harness, not product: the round-1 Python choreography of the distributed construction (dist.py over
comm.py's LoopbackWorld / TorchComm) driving the per-step device ops of libpsacx.so (psacx_op_*, dist_ops.py) or
their numpy twins (tests/numpy_ops.py).  The engine that ships is the C++ one behind psacx_multi_*
(psac_amd/csrc/multi.hpp, psac_amd/multi.py); this package stays because it runs the step ops one by one against
the oracle, on the CPU with real gloo processes and on the GPU."""
