#!/usr/bin/env python
"""Command line of the MI355X VoiceFixer path: ``python -m voicefixer_amd`` == the reference's ``voicefixer`` console
script (voicefixer/__main__.py:69-215): same flags (``-i/-o/-ifdr/-ofdr/--mode/--disable-cuda/--silent/
--weight_prepare``), same argument checks and messages, same ``<name>-mode<k><ext>`` naming for ``--mode all``.

Differences, all deliberate:
  * folder mode (``--infolder``) does not loop files at B = 1 (``__main__.py:187-212``) but hands the folder to
    ``VoiceFixer.restore_folder``: length-sorted ragged batches, decode || restore || encode pipelined;
  * ``--disable-cuda`` cannot move compute to the CPU (there is no CPU implementation in this package): it is accepted and
    only selects host tensors at the API boundary, exactly like ``cuda=False`` everywhere else in voicefixer_amd;
  * mode 2 (train-mode BatchNorm + Dropout) is not built: ``--mode 2`` raises NotImplementedError, ``--mode all`` writes
    modes 0 and 1 and says that mode 2 was skipped;
  * ``--weight_prepare`` cannot download (no network): it only reports whether both checkpoints are in place;
  * ``--gpus N`` (extension, folder mode): the folder is sharded over N MI355X, one process per GPU -- the command
    re-executes itself under ``torch.distributed.run`` on 127.0.0.1 (N clamped, loudly, to the visible devices), every
    rank lists the folder, takes the files ``dist.deal_files`` deals it and writes their outputs; the only collective is
    one all-gather of per-rank counters at the end (RCCL);
  * ``--selfcheck [IN.wav]`` (extension): runs the input through the path with the default (Winograd), the direct and the
    opt-in bf16x3 arithmetic and compares them stage by stage (voicefixer_amd/selfcheck.py) -- the one-command check for
    users with real checkpoints;
  * output formats are WAV and FLAC (``audio_io.FORMATS``) instead of whatever libsndfile offers;
  * folder mode isolates faults per FILE: an unreadable / truncated / too short input or a row the device refuses costs that
    file only -- it is listed on stderr with its reason, every other file is written, the exit status is 2 (all ranks of a
    ``--gpus N`` job learn the list through one all-gather and none of them hangs); ``--skip-existing`` resumes a job.
"""
import argparse
import os
import re
import sys
import time

MODES_BUILT = (0, 1)


def check_output_format(outfile):
    """voicefixer/__main__.py:30-33 with soundfile.available_formats() replaced by what audio_io can write."""
    from . import audio_io
    fmt = re.search(r"\.(\w+)$", outfile)
    assert fmt is not None, "Error: A file-extension for the outfile is missing."
    assert "." + fmt.groups()[0].lower() in audio_io.FORMATS, "Error: Unsupported output format."


def check_arguments(args):
    """voicefixer/__main__.py:36-66."""
    process_file, process_folder = len(args.infile) != 0, len(args.infolder) != 0
    assert process_file or process_folder, (
        "Error: You need to specify a input file path (--infile) or a input folder path (--infolder) to proceed. "
        "For more information please run: voicefixer -h")
    if process_file:
        assert os.path.exists(args.infile), "Error: The input file %s is not found." % args.infile
        output_dirname = os.path.dirname(args.outfile)
        if len(output_dirname) > 1:
            os.makedirs(output_dirname, exist_ok=True)
        check_output_format(args.outfile)
    if process_folder:
        assert os.path.exists(args.infolder), "Error: The input folder %s is not found." % args.infolder
        if len(args.outfolder) > 1:
            os.makedirs(args.outfolder, exist_ok=True)
    return process_file, process_folder


def mode_outfile(outfile, mode, append_mode):
    """voicefixer/__main__.py:13-18: ``<dir>/<base>-mode<k><ext>`` when several modes write next to each other."""
    if not append_mode:
        return outfile
    base, ext = os.path.splitext(os.path.basename(outfile))
    return os.path.join(os.path.dirname(outfile), "{}-mode{}{}".format(base, mode, ext))


def writefile(voicefixer, infile, outfile, mode, append_mode, cuda, verbose=False):
    outfile = mode_outfile(outfile, mode, append_mode)
    if verbose:
        print("Processing {}, mode={}".format(infile, mode))
    start = time.time()
    voicefixer.restore(input=infile, output=outfile, cuda=cuda, mode=int(mode))
    print("Restoration took {} s".format(round(time.time() - start, 1)))


def build_parser():
    parser = argparse.ArgumentParser(prog="voicefixer_amd", description="VoiceFixer - restores degraded speech (MI355X path)")
    parser.add_argument("-i", "--infile", type=str, default="", help="An input file to be processed by VoiceFixer.")
    parser.add_argument("-o", "--outfile", type=str, default="outfile.wav", help="An output file to store the result.")
    parser.add_argument("-ifdr", "--infolder", type=str, default="",
                        help="Input folder. Place all your wav file that need process in this folder.")
    parser.add_argument("-ofdr", "--outfolder", type=str, default="outfolder",
                        help="Output folder. The processed files will be stored in this folder.")
    parser.add_argument("--mode", choices=["0", "1", "2", "all"], default="0",
                        help="0: Original Model (default), 1: Add preprocessing module (remove higher frequencies), "
                             "2: Train mode (not built in this package), all: one output per built mode (0 and 1).")
    parser.add_argument("--disable-cuda", default=False, action="store_true",
                        help="Accepted for compatibility: compute always runs on the MI355X, results are handed over on the host.")
    parser.add_argument("--silent", default=False, action="store_true",
                        help="Set this flag if you do not want to see any message.")
    parser.add_argument("--weight_prepare", default=False, action="store_true",
                        help="Only check that both checkpoints are in place (no network here: nothing is downloaded).")
    parser.add_argument("--batch-size", type=int, default=32, help="(extension) files per ragged batch in folder mode")
    parser.add_argument("--gpus", type=int, default=1,
                        help="(extension) folder mode: shard the folder over this many GPUs (one process per GPU, self-launched)")
    parser.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                        help="(extension) torch.distributed backend of the per-rank counter exchange (nccl == RCCL)")
    parser.add_argument("--skip-existing", default=False, action="store_true",
                        help="(extension) folder mode: leave files whose output already exists alone (resume an interrupted job; "
                             "outputs are written under a temporary name and renamed, so an existing output is a complete one)")
    parser.add_argument("--io-threads", type=int, default=0,
                        help="(extension) folder mode: decode / encode workers per rank (default: host cores / (2 * ranks), 2..8)")
    return parser


def folder_ranks(args, argv):
    """``--gpus N`` given to a PLAIN invocation (no WORLD_SIZE): re-execute under torch.distributed.run, one rank per
    visible device (at most N).  Returns only when a single process is what should run."""
    import torch
    from . import dist as vdist
    visible = torch.cuda.device_count()
    nproc = max(1, min(args.gpus, visible))
    if nproc < args.gpus:
        print("voicefixer_amd: WARNING: --gpus %d requested but only %d HIP device(s) visible -> %d rank(s)"
              % (args.gpus, visible, nproc), file=sys.stderr, flush=True)
    if nproc > 1:
        vdist.exec_ranks(nproc, ["-m", "voicefixer_amd"] + list(argv))


def report_failures(failed, skipped, silent):
    """The per-file outcome of a folder job that the counters do not carry: files that were given up on (ALWAYS printed, to
    stderr: the exit status 2 needs its reasons) and, unless ``--silent``, how many existing outputs ``--skip-existing`` left alone."""
    if skipped and not silent:
        print("skipped %d file(s) whose output already exists" % len(skipped))
    for name, why in failed:
        print("voicefixer_amd: FAILED %s: %s" % (name, why), file=sys.stderr, flush=True)
    if failed:
        print("voicefixer_amd: %d file(s) failed; every other file was written" % len(failed), file=sys.stderr, flush=True)


def report_ranks(per_rank, silent):
    """Rank 0's summary of a sharded folder job (what every rank's ``stats`` dict held, all-gathered)."""
    if silent:
        return
    for r, (files, audio_s, wall_s, dec, enc, stall) in enumerate(per_rank):
        print("rank %d: %d files, %.1f s of audio in %.2f s (%.0fx real time); decode %.2f / encode %.2f worker-seconds, "
              "device waited %.2f s for input" % (r, files, audio_s, wall_s, audio_s / max(wall_s, 1e-9), dec, enc, stall))
    wall = max(x[2] for x in per_rank)
    total = sum(x[1] for x in per_rank)
    print("whole job: %d files, %.1f s of audio in %.2f s = %.0fx real time on %d GPU(s)"
          % (sum(int(x[0]) for x in per_rank), total, wall, total / max(wall, 1e-9), len(per_rank)))


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    if "--selfcheck" in argv:      # (extension) default vs direct vs bf16x3 arithmetic on the loaded checkpoint, stage by stage
        from . import selfcheck
        i = argv.index("--selfcheck")
        return selfcheck.main(argv[:i] + argv[i + 1:])
    args = build_parser().parse_args(argv)
    if args.weight_prepare:
        from . import api
        home = os.path.expanduser("~")
        missing = [p for p in (api.ANALYSIS_CKPT, api.VOCODER_CKPT) if not os.path.exists(os.path.join(home, p))]
        if missing and not args.silent:
            print("Missing checkpoint(s) under ~: %s (no network in this build: place the Zenodo files there)" % ", ".join(missing))
        return 1 if missing else 0
    process_file, process_folder = check_arguments(args)
    if process_file:
        audioext = os.path.splitext(os.path.basename(args.infile))[-1]
        if audioext.lower() not in (".wav", ".flac"):   # (the reference accepts .wav only; FLAC is what its own test reads)
            raise ValueError("Error: Error processing the input file. We only support the .wav format currently. "
                             "Please convert your %s format to .wav. Thanks." % audioext)
    if args.mode == "2":
        raise NotImplementedError("mode 2 (train-mode BatchNorm + Dropout, voicefixer/base.py:114-115) is nondeterministic "
                                  "and not built in voicefixer_amd; modes 0 and 1 are")
    import torch
    from . import api
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and process_folder and "WORLD_SIZE" not in os.environ:
        folder_ranks(args, argv)      # does not return when it re-executes under torch.distributed.run
    rank = 0
    launched = "WORLD_SIZE" in os.environ     # started by torch.distributed.run (our own --gpus N launch, or the user's)
    if launched:
        import torch.distributed as tdist
        rank = int(os.environ.get("RANK", "0"))
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
        from . import dist as vdist
        # (before the process group: the threads gloo / RCCL create inherit the rank's core slice)
        vdist.pin_rank_cpus(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", world)))
        tdist.init_process_group(backend=args.dist_backend)
        if rank != 0:
            args.silent = True        # rank 0 speaks for the job
            process_file = False      # a single file is one rank's work
    cuda = bool(torch.cuda.is_available() and not args.disable_cuda)
    if not args.silent:
        print("Initializing VoiceFixer")
    voicefixer = api.VoiceFixer()
    if not args.silent:
        print("Start processing the input file %s." % args.infile)
    modes = list(MODES_BUILT) if args.mode == "all" else [int(args.mode)]
    append = args.mode == "all"
    if append and not args.silent:
        print("--mode all: writing modes 0 and 1 (mode 2 is not built in this package)")
    if process_file:
        for m in modes:
            writefile(voicefixer, args.infile, args.outfile, m, append, cuda, verbose=not args.silent)
    n_failed = 0
    if process_folder:
        n_files = len([f for f in os.listdir(args.infolder) if os.path.splitext(os.path.basename(f))[-1] == ".wav"])
        if not args.silent:
            print("Found %s .wav files in the input folder %s. Start processing." % (n_files, args.infolder))
        from . import dist as vdist
        for m in modes:
            start = time.time()
            st = {}
            try:
                voicefixer.restore_folder(args.infolder, args.outfolder, mode=m, batch_size=args.batch_size,
                                          name_suffix="-mode%d" % m if append else "", stats=st,
                                          skip_existing=args.skip_existing, io_threads=args.io_threads or None)
            except Exception as e:    # noqa: BLE001 -- per-file faults never get here (restore_folder isolates them); whatever does
                # must not leave the other ranks waiting in the collectives below: this rank reports itself and goes on to them
                import traceback
                traceback.print_exc()
                st.setdefault("failed", []).append(("<rank %d>" % rank, "%s: %s" % (type(e).__name__, e)))
            for k in ("files", "audio_s", "wall_s", "decode_worker_s", "encode_worker_s", "device_waited_for_decode_s"):
                st.setdefault(k, 0.0)
            failed, skipped = list(st.get("failed", [])), list(st.get("skipped", []))
            if launched:
                dev = torch.device("cuda", torch.cuda.current_device()) if args.dist_backend == "nccl" else None
                per_rank = vdist.gather_counters([st["files"], st["audio_s"], st["wall_s"], st["decode_worker_s"],
                                                  st["encode_worker_s"], st["device_waited_for_decode_s"]], dev)
                every = vdist.gather_objects((failed, skipped))      # names and reasons: every rank learns the job's outcome
                failed = sorted(f for fs, _ in every for f in fs)
                skipped = sorted(n for _, sk in every for n in sk)
                if rank == 0:
                    report_ranks(per_rank, args.silent)
            if rank == 0:
                report_failures(failed, skipped, args.silent)
            n_failed += len(failed)
            if not args.silent:
                print("Restoration of %d files (mode %d) took %s s" % (n_files - len(failed) - len(skipped), m, round(time.time() - start, 1)))
    if launched:
        import torch.distributed as tdist
        tdist.barrier()
        tdist.destroy_process_group()
    if not args.silent:
        print("Done")
    return 2 if n_failed else 0      # 2: the job ran to its end, every good file is written, some files were given up on (listed on stderr)


if __name__ == "__main__":
    sys.exit(main())
