"""``python -m voicefixer_amd``: the argument surface and checks of the reference's console script
(voicefixer/__main__.py:30-128), exercised without a device (everything here fails or returns before a model is built)."""
import os

import pytest

from voicefixer_amd import __main__ as cli


def _args(*argv):
    return cli.build_parser().parse_args(list(argv))


def test_flags_and_defaults_are_the_references():
    a = _args()
    assert (a.infile, a.outfile, a.infolder, a.outfolder, a.mode) == ("", "outfile.wav", "", "outfolder", "0")
    assert a.disable_cuda is False and a.silent is False and a.weight_prepare is False
    a = _args("-i", "x.wav", "-o", "y.flac", "-ifdr", "in", "-ofdr", "out", "--mode", "all", "--disable-cuda", "--silent")
    assert (a.infile, a.outfile, a.infolder, a.outfolder, a.mode) == ("x.wav", "y.flac", "in", "out", "all")
    assert a.disable_cuda and a.silent
    with pytest.raises(SystemExit):
        _args("--mode", "3")


def test_check_arguments_messages(tmp_path):
    with pytest.raises(AssertionError, match="You need to specify a input file path"):
        cli.check_arguments(_args())
    with pytest.raises(AssertionError, match="is not found"):
        cli.check_arguments(_args("-i", str(tmp_path / "missing.wav")))
    f = tmp_path / "a.wav"
    f.write_bytes(b"x")
    with pytest.raises(AssertionError, match="file-extension for the outfile is missing"):
        cli.check_arguments(_args("-i", str(f), "-o", str(tmp_path / "noext")))
    with pytest.raises(AssertionError, match="Unsupported output format"):
        cli.check_arguments(_args("-i", str(f), "-o", str(tmp_path / "o.mp3")))
    out = tmp_path / "deep" / "dir" / "o.flac"
    assert cli.check_arguments(_args("-i", str(f), "-o", str(out))) == (True, False)
    assert os.path.isdir(out.parent)                       # the output directory is created (__main__.py:52-54)
    with pytest.raises(AssertionError, match="input folder .* is not found"):
        cli.check_arguments(_args("-ifdr", str(tmp_path / "nofolder")))
    ofd = tmp_path / "outf"
    assert cli.check_arguments(_args("-ifdr", str(tmp_path), "-ofdr", str(ofd))) == (False, True)
    assert os.path.isdir(ofd)


def test_mode_outfile_naming():
    assert cli.mode_outfile("d/out.wav", 1, True) == os.path.join("d", "out-mode1.wav")   # __main__.py:13-18
    assert cli.mode_outfile("d/out.wav", 1, False) == "d/out.wav"


def test_main_rejects_before_building_a_model(tmp_path):
    f = tmp_path / "a.mp3"
    f.write_bytes(b"x")
    with pytest.raises(ValueError, match="only support the .wav format"):
        cli.main(["-i", str(f), "-o", str(tmp_path / "o.wav")])
    w = tmp_path / "a.wav"
    w.write_bytes(b"x")
    with pytest.raises(NotImplementedError, match="mode 2"):
        cli.main(["-i", str(w), "-o", str(tmp_path / "o.wav"), "--mode", "2"])


def test_weight_prepare_reports_missing_checkpoints(tmp_path, monkeypatch, capsys):
    monkeypatch.setenv("HOME", str(tmp_path))
    assert cli.main(["--weight_prepare"]) == 1
    assert "Missing checkpoint" in capsys.readouterr().out
    from voicefixer_amd import api
    for p in (api.ANALYSIS_CKPT, api.VOCODER_CKPT):
        os.makedirs(os.path.dirname(tmp_path / p), exist_ok=True)
        (tmp_path / p).write_bytes(b"")
    assert cli.main(["--weight_prepare", "--silent"]) == 0
