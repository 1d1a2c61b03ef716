"""CPU oracle — TEST INFRASTRUCTURE ONLY.

Plain-PyTorch fp32 restatement of the reference's one-step inference path.  Nothing under
genpercept_amd/ imports this package; only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg do, and only as the checker / reported baseline.
"""
