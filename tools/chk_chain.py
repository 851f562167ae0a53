import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pic1dp_amd as amd
e = amd.Pic1dp(amd.make_input(nparticle_max=100000, nx=1024))
print("chain_mfma layout:", e.kernel_stats(9)[1])
