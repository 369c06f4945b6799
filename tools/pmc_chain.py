"""A few launches of every fused chain kernel at the bench shapes (for rocprofv3 --pmc / --kernel-trace passes).
usage: python3 tools/pmc_chain.py [planes=3] [reps=3]"""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import tools.experiments.check_chain_bwd as cb

planes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cb.run(4096 * 128, 128, planes, reps=reps)
