"""The VCF-only mode of the converter (reference parsing_scripts/parse_ballermix_input.py:55-122) against the reference's own
Example 3 files: input VCF, sample list and expected output are byte-for-byte copies of parsing_scripts/ (data, not code)."""
import filecmp
import gzip
import os

import pytest

from util import GOLD

from ballermixplus_amd import vcf2input

D = os.path.join(GOLD, 'ref_parsing')
VCF = os.path.join(D, 'Example3_first2000var.chr22.phase3_shapeit2_mvncall_integrated_v5b.20130502.genotypes.vcf.gz')
IDS = os.path.join(D, 'Example3_YRI_samples_1KG-v3.20130502.txt')
WANT = os.path.join(D, 'Example3_vcf-only_rec1.25e-6_b0maf-ready.txt')


def test_example3_vcf_only_is_byte_identical(tmp_path, capsys):
    out = tmp_path / 'ex3.txt'
    vcf2input.main(['--vcf', VCF, '-c', '22', '--ID_list', IDS, '--rec_rate', '1.25e-6', '-o', str(out)])     # the readme's command
    assert filecmp.cmp(str(out), WANT, shallow=False)
    assert 'Data from 108 samples will be counted.' in capsys.readouterr().out
    # the output is what the scan's reader takes for B_0,MAF: 376 sites, n = 216, minor counts
    from ballermixplus_amd.hostmodel import InputData
    d = InputData(str(out), MAF=True, nosub=True, phys=True)
    assert d.numSites == 376 and d.sampSizes == {216} and 1 <= d.count.min() and d.count.max() <= 108


def test_vcf_only_edge_cases(tmp_path):
    head = '##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tA\tB\tC\n'
    rec = lambda ch, pos, ref, alt, flt, *gts: '\t'.join([ch, str(pos), '.', ref, alt, '.', flt, '.', 'DP:GT'] + ['7:' + g for g in gts]) + '\n'
    body = (rec('chr7', 100, 'A', 'G', 'PASS', '0|1', '1|1', '0|0') +        # x = 3 of 6 -> minor 3
            rec('7', 200, 'A', 'G', 'PASS', '1|1', '1|1', '0|1') +           # x = 5 of 6 -> minor 1
            rec('7', 300, 'A', 'GT', 'PASS', '0|1', '0|0', '0|0') +          # indel: skipped
            rec('7', 400, 'A', 'G', 'q10', '0|1', '0|0', '0|0') +            # filtered: skipped
            rec('8', 500, 'A', 'G', 'PASS', '0|1', '0|0', '0|0') +           # another chromosome
            rec('7', 600, 'C', 'T', 'PASS', '.|.', '1', '0|0') +             # missing call + a haploid call: x = 1 of 3
            rec('7', 700, 'C', 'T', 'PASS', '0|0', '0|0', '0|0') +           # monomorphic among the samples: skipped
            rec('7', 800, 'C', 'T', 'PASS', '1|1', '1|1', '1|1'))            # fixed: skipped
    plain = tmp_path / 'a.vcf'
    plain.write_text(head + body)
    out = tmp_path / 'o.txt'
    assert vcf2input.convert_vcf_only('7', str(plain), 1e-6, str(out)) == 3
    assert out.read_text().splitlines() == ['position\tgenPos\tx\tn', '100\t9.999999999999999e-05\t3\t6', '200\t0.00019999999999999998\t1\t6',
                                            '600\t0.0006\t1\t3']
    gz = tmp_path / 'a.vcf.gz'
    with gzip.open(gz, 'wt') as f:
        f.write(head + body)
    ids = tmp_path / 'ids.txt'
    ids.write_text('C,A\n')
    assert vcf2input.convert_vcf_only('7', str(gz), 2e-6, str(out), str(ids)) == 2        # samples A and C only
    assert out.read_text().splitlines()[1:] == ['100\t%r\t1\t4' % (100.0 * 2e-6), '200\t%r\t1\t4' % (200.0 * 2e-6)]
    with pytest.raises(SystemExit):
        vcf2input.convert_vcf_only('7', str(tmp_path / 'a.bcf'), 1e-6, str(out))
    tri = tmp_path / 'b.vcf'
    tri.write_text(head + rec('7', 100, 'A', 'G', 'PASS', '0|2', '0|0', '0|0'))
    with pytest.raises(SystemExit):
        vcf2input.convert_vcf_only('7', str(tri), 1e-6, str(out))
    with pytest.raises(SystemExit):
        vcf2input.main(['--vcf', str(plain), '-c', '7', '-o', str(out), '--axt', 'x.axt'])
