import sys, os, warnings
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "style-restricted_gan_amd"))
import torch, bench, collections, traceback
sg = bench.build_trainer(128, 32, 5, torch.device("cuda"))
def batch(s):
    x, src, tgt = bench.synthetic_batch(32, 128, 4, seed=s)
    return x.cuda(), {"source": src.cuda(), "target": tgt}
b = [batch(0), batch(1)]
for s in range(2): sg.train(*b[s])
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
cnt = collections.Counter()
def showwarning(message, category, filename, lineno, file=None, line=None):
    st = traceback.extract_stack()
    fr = [f for f in st if "srgan_amd" in f.filename or "bench" in f.filename]
    key = (fr[-1].filename.split("/")[-1], fr[-1].lineno) if fr else (filename, lineno)
    cnt[key] += 1
warnings.showwarning = showwarning
warnings.simplefilter("always")
sg.train(*b[0])
torch.cuda.set_sync_debug_mode("default")
for k, v in cnt.most_common(20): print(v, k)
