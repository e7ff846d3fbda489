"""Host-side profile (cProfile) of bench.py --config 3 as one rank: where the host spends a DA flow step."""
import argparse, cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("LOCAL_RANK", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import bench
args = argparse.Namespace(gpus=1, steps=30, warmup=3, config=3, chains=8192, no_cpu_baseline=True, dry_run=False)
pr = cProfile.Profile(); pr.enable()
bench.run_rank(args)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats("_batched|hmcda|_plugin|bench.py")
st.sort_stats("tottime").print_stats(12)
