"""Sanity run (GPU box): 300 optimisation steps on ONE fixed LJSpeech-shaped batch of 8 utterances at production dims - the
loss must fall steadily (Adam + clip + L2 as in model/tts_model.py:78-91), a quick end-to-end check that the gradients train."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import VANILLA
from tacotron2_amd.init import init_parameters
from tacotron2_amd.params import ParamStore
from tacotron2_amd.synthetic import ljspeech_batch
from tacotron2_amd.trainer import Trainer
dev = torch.device("cuda:0")
ps = ParamStore(VANILLA, dev); init_parameters(ps, 0)
tr = Trainer(ps, lr=1e-3, weight_decay=1e-6)
batch = {k: v.to(dev) for k, v in ljspeech_batch(8, seed=7, num_speakers=4).items()}
t0 = time.time()
for step in range(300):
    loss3, _ = tr.train_step(batch)
    if step % 25 == 0 or step == 299:
        l = [float(x) for x in loss3.cpu()]
        print(f"step {step:3d}: gate {l[0]:.4f} mel {l[1]:.4f} post {l[2]:.4f} total {sum(l):.4f}  ({time.time() - t0:.1f} s)", flush=True)
tr.engine.check_persistent_kernels()
