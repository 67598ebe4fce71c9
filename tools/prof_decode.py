"""Autoregressive decode only (for rocprofv3): 64 utterances, 200 frames, stop checks live."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import VANILLA
from tacotron2_amd.engine import Engine
from tacotron2_amd.init import init_parameters
from tacotron2_amd.params import ParamStore
from tacotron2_amd.synthetic import ljspeech_batch
dev = torch.device("cuda:0")
ps = ParamStore(VANILLA, dev); init_parameters(ps, 0)
eng = Engine(ps)
ib = ljspeech_batch(64, seed=4321, num_speakers=4)
ci, cl, spk = ib["chars_idx"].to(dev), ib["chars_idx_len"].to(dev), ib["speaker_id"].to(dev)
eng.infer(ci, cl, 32, speaker_id=spk, training=False, seed=1)
eng.infer(ci, cl, 200, speaker_id=spk, training=False, seed=2, check_every=64)
torch.cuda.synchronize()
