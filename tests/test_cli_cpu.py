"""The drop-in command line without a GPU: every flag of the reference's parser is accepted with
the same spelling/defaults (BalLeRMix+_v1.py:718-753), the helper-file step runs end to end, and a
scan without a device fails loudly instead of falling back to anything."""
import os
import subprocess
import sys

import pytest

from util import GOLD, REFT, REPO

from ballermixplus_amd.cli import build_parser

REF_FLAGS = ['-i', '--input', '-o', '--output', '--spect', '--minCount', '--getSpect', '--getConfig', '--findBal',
             '--findPos', '--noFreq', '--noSub', '--MAF', '--usePhysPos', '--rec', '--fixWinSize', '-w', '--window',
             '--noCenter', '-s', '--step', '--fixX', '--fixAlpha', '--rangeA', '--listA']


def test_every_reference_flag_is_accepted_with_the_reference_defaults():
    p = build_parser()
    known = set()
    for a in p._actions:
        known.update(a.option_strings)
    assert not [f for f in REF_FLAGS if f not in known]
    o = p.parse_args(['-i', 'x', '--spect', 'y'])
    assert (o.minCount, o.getSpec, o.getConfig, o.bal, o.pos, o.nofreq, o.nosub, o.MAF, o.phys) == \
        (1, False, False, False, False, False, False, False, False)
    assert (o.Rrate, o.size, o.w, o.noCenter, o.step, o.x, o.abeta, o.seqA, o.listA, o.outfile) == \
        (1e-6, False, 0, False, 1, None, None, None, None, None)
    o = p.parse_args(['-i', 'x', '--spect', 'y', '-s', '25', '-w', '50', '--fixAlpha', '7', '--rec', '2e-6'])
    assert o.step == 25.0 and isinstance(o.step, float) and o.w == 50 and o.abeta == 7.0 and o.Rrate == 2e-6
    with pytest.raises(SystemExit):
        p.parse_args(['-i', 'x'])                     # --spect is required, as in the reference


def test_cli_helper_file_step_end_to_end(tmp_path):
    out = tmp_path / 'spect.txt'
    r = subprocess.run([sys.executable, os.path.join(REPO, 'BalLeRMixPlus_amd.py'), '-i',
                        os.path.join(REFT, 'Example2_balancing_10MYA_MAF.txt'), '--spect', str(out), '--getSpect', '--MAF'],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and 'Done.' in r.stdout
    assert out.read_bytes() == open(os.path.join(GOLD, 'helpers', 'spect_ex2_MAF.txt'), 'rb').read()


def test_cli_scan_without_a_gpu_fails_loudly(tmp_path):
    from ballermixplus_amd import _lib
    if _lib.lib().bmx_device_count() > 0:
        pytest.skip('a GPU is present')
    r = subprocess.run([sys.executable, os.path.join(REPO, 'BalLeRMixPlus_amd.py'), '-i',
                        os.path.join(REFT, 'Example1_fullSweep_200kya_DAF.txt'), '--spect',
                        os.path.join(REFT, 'HC_CEU_Neut_DAF_spect_for_B2.txt'), '-o', str(tmp_path / 'o.txt')],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'no CPU fallback' in (r.stderr + r.stdout)
    assert not (tmp_path / 'o.txt').exists()


def test_input_list_and_shard_block_helpers(tmp_path, monkeypatch, capsys):
    """--inputs list files (comments and blank lines wherever they are indented, relative names against the list's own directory)
    and BMX_SHARD_BLOCK (a positive multiple of 16, checked for the one-file and the many-file form alike)."""
    from ballermixplus_amd import cli
    sub = tmp_path / 'lists'
    sub.mkdir()
    lst = sub / 'genome.txt'
    lst.write_text('# whole genome\nchr1.txt\n   # an indented comment\n\n  sub/chr2.txt  \n/abs/chr3.txt\n')
    assert cli.input_list(str(lst)) == [str(sub / 'chr1.txt'), str(sub / 'sub/chr2.txt'), '/abs/chr3.txt']
    monkeypatch.delenv('BMX_SHARD_BLOCK', raising=False)
    assert cli.shard_block() is None
    monkeypatch.setenv('BMX_SHARD_BLOCK', '64')
    assert cli.shard_block() == 64
    for bad in ('0', '-16', '24', 'many'):
        monkeypatch.setenv('BMX_SHARD_BLOCK', bad)
        with pytest.raises(SystemExit):
            cli.shard_block()
    assert 'multiple of 16' in capsys.readouterr().out
    # --inputs together with a helper-file step is refused with a message (it used to fall through with infile = None)
    with pytest.raises(SystemExit):
        cli.main(['--inputs', str(lst), '--spect', str(tmp_path / 's.txt'), '--getSpect'])
    assert '--inputs lists files to scan' in capsys.readouterr().out
    assert cli.output_name(str(tmp_path / 'out_{}.tsv'), '/x/chr7.txt') == str(tmp_path / 'out_chr7.tsv')
