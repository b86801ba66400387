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
