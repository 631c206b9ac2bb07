"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the GIST hot path (GraphSAGE forward/backward over cluster
sub-graphs, cluster batch extraction, IST partition/dispatch/sync), used as the
parity checker for the HIP path and as the reported CPU baseline.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
import this package.  The product package `gist_amd` never does.

Pinning: the restatement is checked against golden vectors recorded from the
reference's own code (tests/golden/*.npz, produced by oracle/gen_golden.py which
imports /root/reference against the DGL stub in oracle/dgl_stub/).
"""
