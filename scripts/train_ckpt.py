"""Train the synthetic checkpoint on the GPU box and save it in the reference's schema (for bench.py --checkpoint).
usage: train_ckpt.py [easy|hard] [steps] [out.pth]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd  # noqa: F401
from tools.train_synth import train_checkpoint
task = sys.argv[1] if len(sys.argv) > 1 else 'easy'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
out = sys.argv[3] if len(sys.argv) > 3 else f'/tmp/vfn_trained_{task}.pth'
sd, info = train_checkpoint(torch.device('cuda', 0), steps=steps, task=task, log=lambda m: print(m, flush=True) if 'step' in m and int(m.split()[1].rstrip(':')) % 500 == 0 else None)
print(info, flush=True)
torch.save({'epoch': 0, 'model': sd, 'loss': info['loss_last_50'], 'seed': 20200212, 'task': task}, out)
print('saved', out)
