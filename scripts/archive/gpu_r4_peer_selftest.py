"""round 4: sns_peer_selftest (in-process protocol self-test / latency probe of the peer-window transport) from a process that does
nothing else.  usage: python scripts/gpu_r4_peer_selftest.py NRANKS HALO_NODES [REPS]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabilized_navier_stokes_flow_fenicsx_amd import _lib
lib = _lib.load()
n, halo = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
us = (C.c_double * 3)()
rc = lib.sns_peer_selftest(0, n, halo, reps, us)
print(f"{n} ranks, {halo} halo nodes ({32 * halo / 1e3:.0f} kB per link), {reps} rounds: rc {rc} "
      + (f"exchange {us[0]:.1f} us, all-reduce {us[1]:.1f} us, all-gather {us[2]:.1f} us per round (the collective alone, back to back)"
         if rc == 0 else lib.sns_last_error().decode()), flush=True)
sys.exit(0 if rc == 0 else 1)
