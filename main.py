#!/usr/bin/env python3
"""CLI with the reference's surface for the hot path (main.py:15-107,150-226 of mattm458/tacotron2):

    python main.py --config C --device N train --speech-dir S [--results-dir R] [--resume-ckpt K] [--finetune --finetune-steps n]
    python main.py --config C --device N say --checkpoint K --text "..." [--out out.npy] [--random-seed s] [--speaker-id i]
    python main.py --config C --device N test --speech-dir S --checkpoint K [--hifi-gan-checkpoint G] [--results-dir R]
    python main.py --config C --device N test-correlation --speech-dir S --checkpoint K [--hifi-gan-checkpoint G] [--results-dir R]
    python main.py --config C --device N train-mel-export --speech-dir S --checkpoint K [--results-dir R]

Other reference sub-commands (preprocess, server) are data preparation / demo tooling outside the hot-path scope
(SURVEY.md section 2).  Multi-GPU training: `python -m torch.distributed.run --nproc-per-node N
main.py --config C train ...` (one process per GPU, RCCL gradient all-reduce)."""
import click

from tacotron2_amd.run.common import load_config


@click.group()
@click.pass_context
@click.option("--config", type=str, required=False, default=None, help="A Tacotron hyperparameter config file")
@click.option("--device", type=int, required=False, default=0, help="The GPU to use for training or inference. Default 0.")
def main(ctx, config, device):
    ctx.ensure_object(dict)
    ctx.obj["config"] = load_config(config) if config is not None else None
    ctx.obj["device"] = device


@main.command()
@click.pass_context
@click.option("--speech-dir", required=True, type=str, help="A directory containing audio files from the dataset.")
@click.option("--results-dir", required=False, type=str, help="The directory to save results.")
@click.option("--resume-ckpt", required=False, type=str, help="Resume training from the given checkpoint.")
@click.option("--prosody-model-checkpoint", required=False, type=str, help="(accepted for CLI compatibility; unused)")
@click.option("--finetune", is_flag=True, default=False, help="Fine-tune a model. If specified, --resume-ckpt is required.")
@click.option("--finetune-steps", required=False, type=int, help="Steps to fine-tune. Required if --finetune is given.")
@click.option("--max-steps", required=False, type=int, default=None, help="Override training.args.max_steps (smoke runs).")
@click.option("--synthetic", is_flag=True, default=False, help="Train on synthetic LJSpeech-shaped batches (no dataset needed).")
def train(ctx, speech_dir, results_dir=None, resume_ckpt=None, prosody_model_checkpoint=None, finetune=False,
          finetune_steps=None, max_steps=None, synthetic=False):
    if ctx.obj["config"] is None:
        raise Exception("Configuration required for training!")
    if finetune and finetune_steps is None:
        raise Exception("If finetuning, --finetune-steps is required!")
    from tacotron2_amd.run.train import do_train
    c = ctx.obj["config"]
    do_train(dataset_config=c["dataset"], training_config=c["training"], model_config=c["model"],
             extensions_config=c["extensions"], device=ctx.obj["device"], speech_dir=speech_dir, results_dir=results_dir,
             resume_ckpt=resume_ckpt, finetune=finetune, finetune_steps=finetune_steps, max_steps_override=max_steps,
             synthetic=synthetic)


@main.command()
@click.pass_context
@click.option("--checkpoint", required=True, type=str, help="A trained Tacotron model checkpoint")
@click.option("--text", required=True, type=str, help="Text to speak")
@click.option("--out", required=False, type=str, default="out.npy", help="Output file (log-mel .npy). Default: out.npy")
@click.option("--hifi-gan-checkpoint", required=False, type=str, default=None, help="HiFi-GAN generator checkpoint (config.json next to it, UNIVERSAL_V1 values when absent)")
@click.option("--random-seed", required=False, type=int, default=None, help="A random seed to use in generation.")
@click.option("--speaker-id", required=False, type=int, default=None, help="Speaker ID for a multi-speaker model")
@click.option("--controls", required=False, type=str, default=None, help="If controls are enabled, a comma-separated list of values to pass into the model. Defaults to all 0 values.")
@click.option("--description", required=False, type=str, default=None, help="Path of a precomputed description embedding (.pt / .npy, pooler_output of bert-base-uncased); raw text needs the BERT weights (unavailable offline)")
def say(ctx, checkpoint, text, out, speaker_id, hifi_gan_checkpoint, random_seed, controls, description):
    if ctx.obj["config"] is None:
        raise Exception("Configuration required for speech!")
    from tacotron2_amd.run.say import do_say
    c = ctx.obj["config"]
    do_say(dataset_config=c["dataset"], training_config=c["training"], model_config=c["model"],
           extensions_config=c["extensions"], device=ctx.obj["device"], checkpoint=checkpoint, text=text, output=out,
           speaker_id=speaker_id, hifi_gan_checkpoint=hifi_gan_checkpoint, random_seed=random_seed, controls=controls,
           description=description)


@main.command()
@click.pass_context
@click.option("--speech-dir", required=True, type=str, help="A directory containing audio files from the dataset.")
@click.option("--checkpoint", required=True, type=str, help="A trained Tacotron model checkpoint")
@click.option("--hifi-gan-checkpoint", required=False, type=str, default=None, help="A HiFi-GAN generator checkpoint. If not given, Griffin-Lim is used.")
@click.option("--results-dir", required=False, type=str, default=None, help="The directory to save results.")
@click.option("--batch-size", required=False, type=int, default=8, help="Utterances decoded together (reference: 8; up to 64 per decode group).")
@click.option("--max-len", required=False, type=int, default=5000, help="Frame cap per utterance (reference: 5000).")
@click.option("--limit", required=False, type=int, default=None, help="Only the first n utterances of the test manifest.")
def test(ctx, speech_dir, checkpoint, hifi_gan_checkpoint, results_dir, batch_size, max_len, limit):
    """Synthesise the test manifest (run/test.py of the reference): one wav per utterance + failures.csv."""
    if ctx.obj["config"] is None:
        raise Exception("Configuration required for testing!")
    from tacotron2_amd.run.test import do_test
    c = ctx.obj["config"]
    do_test(dataset_config=c["dataset"], training_config=c["training"], model_config=c["model"],
            extensions_config=c["extensions"], device=ctx.obj["device"], speech_dir=speech_dir, checkpoint=checkpoint,
            hifi_gan_checkpoint=hifi_gan_checkpoint, results_dir=results_dir, batch_size=batch_size, max_len=max_len, limit=limit)


@main.command()
@click.pass_context
@click.option("--speech-dir", required=True, type=str, help="A directory containing audio files from the dataset.")
@click.option("--checkpoint", required=True, type=str, help="A trained Tacotron model checkpoint")
@click.option("--hifi-gan-checkpoint", required=False, type=str, default=None, help="A trained HiFi-GAN model checkpoint")
@click.option("--results-dir", required=False, type=str, default=None, help="The directory to save results.")
@click.option("--samples-per-speaker", required=False, type=int, default=200, help="Utterances drawn per speaker (reference: 200).")
@click.option("--max-len", required=False, type=int, default=5000, help="Frame cap per utterance (reference: 5000).")
@click.option("--limit-overrides", required=False, type=int, default=None, help="Only the first n of the 51 control overrides.")
def test_correlation(ctx, speech_dir, checkpoint, hifi_gan_checkpoint, results_dir, samples_per_speaker, max_len, limit_overrides):
    """The test manifest under 51 control-vector overrides (run/test_correlation.py of the reference)."""
    if ctx.obj["config"] is None:
        raise Exception("Configuration required for testing!")
    from tacotron2_amd.run.test_correlation import do_test_correlation
    c = ctx.obj["config"]
    do_test_correlation(dataset_config=c["dataset"], training_config=c["training"], model_config=c["model"],
                        extensions_config=c["extensions"], device=ctx.obj["device"], speech_dir=speech_dir, checkpoint=checkpoint,
                        hifi_gan_checkpoint=hifi_gan_checkpoint, results_dir=results_dir, samples_per_speaker=samples_per_speaker,
                        max_len=max_len, limit_overrides=limit_overrides)


@main.command()
@click.pass_context
@click.option("--speech-dir", required=True, type=str, help="A directory containing audio files from the dataset.")
@click.option("--checkpoint", required=True, type=str, help="A trained Tacotron model checkpoint")
@click.option("--results-dir", required=False, type=str, default=None, help="The directory to save results. Defaults to the model configuration name with a timestamp.")
def train_mel_export(ctx, speech_dir, checkpoint, results_dir=None):
    """Teacher-forced post-net mels of the train + val manifests (run/train_mel_export.py of the reference)."""
    if ctx.obj["config"] is None:
        raise Exception("Configuration required!")
    from tacotron2_amd.run.train_mel_export import do_train_mel_export
    c = ctx.obj["config"]
    do_train_mel_export(dataset_config=c["dataset"], training_config=c["training"], model_config=c["model"],
                        extensions_config=c["extensions"], device=ctx.obj["device"], speech_dir=speech_dir, checkpoint=checkpoint,
                        results_dir=results_dir)


if __name__ == "__main__":
    main(obj={})
