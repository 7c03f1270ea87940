#!/usr/bin/env python3
"""Command line front end of the MI355X-native segmenter.

Same options as the reference's scripts/ina_speech_segmenter.py (-i -o -s -d -g -b -e -r, :45-53) and the
same output naming (<basename>.<format> inside the output directory, :80-84).  Extra: --models synthetic
(seeded stand-in weights, for machines without the Keras release assets) and multi-GPU operation: start it
under `python -m torch.distributed.run --nproc-per-node N` and the inputs are dealt to the N GPUs.
"""
import argparse
import glob
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _truthy(v):
    v = str(v).strip().lower()
    if v in ('y', 'yes', 't', 'true', 'on', '1'):
        return True
    if v in ('n', 'no', 'f', 'false', 'off', '0'):
        return False
    raise argparse.ArgumentTypeError(f'invalid truth value {v!r}')


def build_parser():
    ap = argparse.ArgumentParser(
        description="Speech/Music(/Noise) and Male/Female segmentation into CSV or Praat TextGrid files. 'noEnergy' segments "
                    "are excluded from the music / noise / speech / gender analysis.")
    ap.add_argument('-i', '--input', nargs='+', required=True, help='media paths, glob patterns or http(s) URLs')
    ap.add_argument('-o', '--output_directory', required=True, help='directory receiving <basename>.<format>')
    ap.add_argument('-s', '--batch_size', type=int, default=32, help='accepted for compatibility (the engine sizes its own passes)')
    ap.add_argument('-d', '--vad_engine', choices=['sm', 'smn'], default='smn')
    ap.add_argument('-g', '--detect_gender', type=_truthy, default=True, metavar='{true,false}')
    ap.add_argument('-b', '--ffmpeg_binary', default='ffmpeg', help="ffmpeg binary; 'None' reads 16 kHz mono WAV directly")
    ap.add_argument('-e', '--export_format', choices=['csv', 'textgrid'], default='csv')
    ap.add_argument('-r', '--energy_ratio', type=float, default=0.03)
    ap.add_argument('--models', default=None, help="'synthetic' = seeded stand-in weights")
    return ap


def main(argv=None):
    args = build_parser().parse_args(argv)
    ffmpeg = None if args.ffmpeg_binary.lower() in ('none', '') else args.ffmpeg_binary
    if ffmpeg is None:
        print('Disabling ffmpeg. Make sure your audio files are already sampled at 16kHz.')
    inputs = []
    for pat in args.input:
        inputs += [pat] if pat.startswith('http') else sorted(glob.glob(pat))
    assert len(inputs) > 0, 'No existing media selected for analysis! Bad values provided to -i (%s)' % args.input
    odir = args.output_directory.strip(' \t\n\r').rstrip('/')
    assert os.access(odir, os.W_OK), 'Directory %s is not writable!' % odir
    outputs = [os.path.join(odir, os.path.splitext(os.path.basename(f))[0] + '.' + args.export_format) for f in inputs]

    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    from inaspeechsegmenter_amd import Segmenter
    seg = Segmenter(vad_engine=args.vad_engine, detect_gender=args.detect_gender, ffmpeg=ffmpeg, batch_size=args.batch_size,
                    energy_ratio=args.energy_ratio, device=local_rank, models=args.models)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        if world == 1:
            seg.batch_process(inputs, outputs, verbose=True, output_format=args.export_format)
            return 0
        import torch
        import torch.distributed as dist
        from inaspeechsegmenter_amd.archive import segment_archive
        torch.cuda.set_device(local_rank)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        table, lmsg = segment_archive(seg, inputs, outputs, output_format=args.export_format)
        if dist.get_rank() == 0:
            print('%d files, %d segments gathered from %d GPUs' % (len(table), sum(map(len, table.values())), world))
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
