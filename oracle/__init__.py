"""CPU oracles -- TEST INFRASTRUCTURE ONLY (see the header of each file).

May be imported from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
only; never from dspnet_amd/."""
