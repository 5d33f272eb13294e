"""Physical constants in Nano-kappa's unit system (angstrom, ps, K, eV, THz*rad).

Mirrors reference classes/Constants.py:5-12, which reads scipy.constants (CODATA 2018;
all three are exact SI-2019 derived values, so they do not drift between scipy versions).
"""
import math


class Constants(object):
    def __init__(self):
        self.hbar = 6.582119569e-16 * 1e12      # eV ps / rad   (Constants.py:7)
        self.kb = 8.617333262e-05               # eV / K        (Constants.py:8)
        self.ev_in_J = 1.602176634e-19          # J / eV        (Constants.py:9)
        self.a_in_m = 1e-10                     # m / angstrom  (Constants.py:10)
        self.ps_in_s = 1e-12                    # s / ps        (Constants.py:11)
        self.eVpsa2_in_Wm2 = self.ev_in_J / (self.ps_in_s * (self.a_in_m) ** 2)   # (Constants.py:12)
        self.pi = math.pi
