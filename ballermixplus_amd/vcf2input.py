"""VCF -> BalLeRMix+ input, the VCF-only mode (reference parsing_scripts/parse_ballermix_input.py:55-122, `parse_vcf_only`;
SURVEY.md section 8(f) row 4, last clause).  Host code, one streaming pass; not on the GPU path.

    python -m ballermixplus_amd.vcf2input --vcf calls.vcf.gz -c 22 [--ID_list ids.txt] [--rec_rate 1e-6] -o out.txt

Same flags, same output format as the reference: header `position genPos x n`, one row per bi-allelic PASS SNP of the
chromosome with the MINOR allele count x among the n called alleles of the chosen samples and genPos = float(POS) * rec_rate
(printed as Python prints the product) -- input for B_0,MAF (`--noSub --MAF`).  With only a VCF there is no outgroup and no
map, so the modes that need `--axt` / `--rec_map` are not part of this build (their alignment fixture is absent from the
reference checkout as well).

One deviation, of the --rangeA kind: the reference as shipped raises `ValueError: Invalid x: 0` at the first
site that is monomorphic among the chosen samples -- on its own Example 3 too -- while its committed expected output
(parsing_scripts/test_output/Example3_vcf-only_rec1.25e-6_b0maf-ready.txt) simply lacks those sites.  This build does what
that file shows: sites with x = 0 or x = n are skipped."""
import argparse
import gzip
import re
import sys
import time

BASES = frozenset('ACGT')
CALLED = re.compile(r'[0-9]+')          # allele indices in a GT string ('0|1', '1/1', '0'); '.' is a missing call


def open_text(path, suffix):
    """.gz through gzip, `suffix` as plain text; anything else is refused with the reference's message."""
    low = path.lower()
    if low.endswith('.gz'):
        return gzip.open(path, 'rt')
    if low.endswith(suffix):
        return open(path, 'r')
    print(f"Unrecognized {suffix.upper()} file name. Please make sure it's in {suffix} or {suffix}.gz format.")
    sys.exit(1)


def sample_columns(header_fields, id_list_file):
    """Column indices of the samples to count: all of them (from column 9 on) or those named in the comma-separated list file."""
    if id_list_file is None:
        cols = list(range(9, len(header_fields)))
    else:
        with open(id_list_file) as f:
            names = f.read().strip().split(',')
        where = {name: j for j, name in enumerate(header_fields)}
        missing = [n for n in names if n not in where]
        if missing:
            raise ValueError('%r is not in list' % missing[0])       # (the reference's header.index() raises the same way)
        cols = sorted(set(where[n] for n in names))
        assert cols[0] >= 9
    print(f'Data from {len(cols)} samples will be counted.')
    return cols


def allele_counts(fields, cols, gt_at):
    """(alternate alleles, called alleles) over the chosen samples of one VCF record."""
    x = n = 0
    for j in cols:
        gt = fields[j].split(':')[gt_at]
        called = CALLED.findall(gt)
        if not called:
            assert '.' in gt
            continue
        if any(a not in ('0', '1') for a in called):
            print('Warning: This script only applies to diploid and haploid data.')
            print(gt, tuple(int(a) for a in called), '')
            sys.exit(1)
        x += called.count('1')
        n += len(called)
    return x, n


def convert_vcf_only(chrom, vcffile, rec_rate, outfile, id_list_file=None):
    """The whole conversion; returns the number of rows written."""
    names = {chrom, 'chr' + chrom}
    rows = 0
    with open_text(vcffile, '.vcf') as vcf, open(outfile, 'w') as out:
        out.write('position\tgenPos\tx\tn\n')
        cols = None
        for line in vcf:
            if line.startswith('##'):
                continue
            f = line.strip().split('\t')
            if cols is None:                         # the #CHROM line
                cols = sample_columns(f, id_list_file)
                continue
            if f[0] not in names or f[3] not in BASES or f[4] not in BASES or 'PASS' not in f[6]:
                continue                             # another chromosome, not a bi-allelic SNP, or filtered
            x, n = allele_counts(f, cols, f[8].split(':').index('GT'))
            if 0 < x < n:
                out.write(f'{f[1]}\t{float(f[1]) * rec_rate}\t{min(x, n - x)}\t{n}\n')
                rows += 1
    return rows


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument('--vcf', dest='vcffile', required=True, help='Path and name of the vcf file (.vcf or .vcf.gz).')
    ap.add_argument('-c', '--chr', dest='ch', required=True, help='ID of the chromosome. E.g. 2a for chr2a, 12 for chr12, etc.')
    ap.add_argument('-o', '--output', dest='outfile', required=True, help='Path and name of the output file.')
    ap.add_argument('--ID_list', dest='pop_list', default=None,
                    help='File with the sample IDs to count (their column names in the vcf), separated by commas. Default: all samples.')
    ap.add_argument('--axt', dest='axtfile', default=None, help='(reference flag; the alignment modes are not part of this build)')
    ap.add_argument('--rec_rate', dest='rec_rate', type=float, default=1e-6, help='Recombination rate in cM/nt. Default 1e-6.')
    ap.add_argument('--rec_map', dest='rec_map', default=None, help='(reference flag; the recombination-map modes are not part of this build)')
    ap.add_argument('--hap', dest='hap', action='store_true', default=False, help='(reference flag; only matters with --axt)')
    return ap


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    ap = build_parser()
    if not argv:
        ap.print_help()
        sys.exit()
    opt = ap.parse_args(argv)
    if opt.axtfile is not None or opt.rec_map is not None:
        print('This build converts VCF-only input (for B_0,MAF); --axt and --rec_map are handled by the reference\'s parsing script.')
        sys.exit(1)
    t0 = time.time()
    print(time.ctime(), f'Parsing BalLeRMix input for B_0maf with a uniform recombination rate of {opt.rec_rate} cM/nt...')
    convert_vcf_only(opt.ch, opt.vcffile, opt.rec_rate, opt.outfile, opt.pop_list)
    print(time.ctime(), f'Parsing completed. Output file: {opt.outfile} Total time {time.time() - t0}sec.')


if __name__ == '__main__':
    main()
