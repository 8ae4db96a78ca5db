import cProfile, pstats, sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/examples")
import trainer_standin
trainer_standin.train("DDPG", cycles=5, verbose=False)   # warm
pr = cProfile.Profile(); pr.enable()
trainer_standin.train("DDPG", cycles=60, verbose=False)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
