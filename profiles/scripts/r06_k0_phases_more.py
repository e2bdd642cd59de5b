import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "profiles", "scripts"))
import r06_generic_time as G
G.REPS = 1
G.dae(20, 10, 40, 40, 64, "euler")
G.dae(8, 4, 6, 6, 64, "rk4")
G.ode(20, 2, (64, 64, 64), "euler")
