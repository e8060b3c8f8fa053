"""`convert`: BAM/CRAM -> per-chromosome int32 bin counts (mirror of the reference's
convert_tools.convert_reads, convert_tools.py:15-120).  I/O-bound htslib iteration, outside the
MI355X hot path; needs pysam (not installed in the build image, so this module is imported lazily
by main.tool_convert).  Same filters and bookkeeping as the reference:
  * contigs 1-22, X, Y (optional "chr" prefix); bins = int(length / binsize + 1);
  * paired reads must be proper pairs; a read is a duplicate when it starts where the previous
    read started (and, for pairs, its mate starts where the previous paired read's mate started)
    unless --normdup; MAPQ >= 1; bin = int(pos / binsize).
The per-read Python branching of the reference is replaced by array operations on the read
columns, contig by contig.
"""
import logging
import sys

import numpy as np


def _open(args):
    import pysam
    if args.infile.endswith(".bam"):
        return pysam.AlignmentFile(args.infile, "rb")
    if args.infile.endswith(".cram"):
        if args.reference is None:
            logging.error("Cram support requires a reference file, please use the --reference argument")
            sys.exit(1)
        return pysam.AlignmentFile(args.infile, "rc", reference_filename=args.reference)
    logging.error("Unsupported input file type. Make sure your input filename has a correct "
                  "extension ( bam or cram)")
    sys.exit(1)


def count_contig(pos, mate, mapq, paired, proper, n_bins, binsize, normdup, last_pos, last_mate):
    """Counts of one contig from its read columns.  last_pos / last_mate carry the previous
    considered read's start and the previous paired read's mate start across contigs
    (the reference's larp / larp2).  Returns (counts, stats dict, last_pos, last_mate)."""
    counts = np.zeros(n_bins, dtype=np.int32)
    stats = {"seen": 0, "dup": 0, "mapq": 0, "pair_fail": 0}
    if len(pos) == 0:
        return counts, stats, last_pos, last_mate
    pos = np.asarray(pos, dtype=np.int64)
    mate = np.asarray(mate, dtype=np.int64)
    mapq = np.asarray(mapq)
    paired = np.asarray(paired, dtype=bool)
    proper = np.asarray(proper, dtype=bool)
    considered = ~paired | proper                     # improper pairs are skipped entirely
    stats["pair_fail"] = int(np.sum(paired & ~proper))
    p, m, q, pr = pos[considered], mate[considered], mapq[considered], paired[considered]
    if len(p) == 0:
        return counts, stats, last_pos, last_mate
    prev_pos = np.concatenate(([last_pos], p[:-1]))
    # mate start of the most recent PAIRED read before each read (forward fill)
    fill = np.where(pr, np.arange(len(p)), -1)
    np.maximum.accumulate(fill, out=fill)
    shifted = np.concatenate(([-1], fill[:-1]))
    prev_mate = np.where(shifted >= 0, m[np.maximum(shifted, 0)], last_mate)
    dup = np.zeros(len(p), dtype=bool)
    if not normdup:
        dup = (prev_pos == p) & (~pr | (prev_mate == m))
    good = ~dup & (q >= 1)
    np.add.at(counts, (p[good] / binsize).astype(np.int64), 1)
    stats.update(seen=len(p), dup=int(dup.sum()), mapq=int(np.sum(~dup & (q < 1))))
    last_pos = int(p[-1])
    if pr.any():
        last_mate = int(m[np.flatnonzero(pr)[-1]])
    return counts, stats, last_pos, last_mate


def convert_reads(args):
    bins_per_chr = {str(c): None for c in range(1, 25)}
    logging.info("Importing data ...")
    reads_file = _open(args)
    tot = {"seen": 0, "dup": 0, "mapq": 0, "pair_fail": 0}
    kept = 0
    last_pos, last_mate = -1, -1
    logging.info("Converting aligned reads ... This might take a while ...")
    for index, contig in enumerate(reads_file.references):
        name = contig[3:] if contig[:3].lower() == "chr" else contig
        if name not in bins_per_chr and name not in ("X", "Y"):
            continue
        n_bins = int(reads_file.lengths[index] / float(args.binsize) + 1)
        logging.info("Working at {}; processing {} bins".format(contig, n_bins))
        cols = [[], [], [], [], []]
        for read in reads_file.fetch(contig):
            cols[0].append(read.pos)
            cols[1].append(read.next_reference_start)
            cols[2].append(read.mapping_quality)
            cols[3].append(read.is_paired)
            cols[4].append(read.is_proper_pair)
        counts, st, last_pos, last_mate = count_contig(*cols, n_bins, args.binsize, args.normdup,
                                                       last_pos, last_mate)
        for key in tot:
            tot[key] += st[key]
        bins_per_chr[{"X": "23", "Y": "24"}.get(name, name)] = counts
        kept += int(counts.sum())
    qual_info = {
        "mapped": reads_file.mapped, "unmapped": reads_file.unmapped,
        "no_coordinate": reads_file.nocoordinate, "filter_rmdup": tot["dup"],
        "filter_mapq": tot["mapq"], "pre_retro": tot["seen"], "post_retro": kept,
        "pair_fail": tot["pair_fail"],
    }
    return bins_per_chr, qual_info
