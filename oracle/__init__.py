"""ORACLE -- test infrastructure only.

CPU restatement of the ModelCrowdNav rollout hot path, used as the checker by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
modelcrowdnav_amd/ may import this package.

  cport.py   ctypes binding of libmcn_oracle.so (mcn_oracle.c: ORCA f32 solve,
             env step, swept-circle test, look-ahead reward)
  pyref.py   numpy / torch-fp32 restatements of the Python-side arithmetic
             (action table, scenario generator, rotate, SARL value net, SGAN step)

Pinning status is written at the top of each file and in DESIGN.md.
"""
