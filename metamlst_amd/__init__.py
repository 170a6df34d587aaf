"""metamlst_amd -- MI355X-native MLST-typing hot path (drop-in for the bowtie2 + samtools +
pysam + cmseq path of SegataLab/metamlst).  See DESIGN.md."""
__version__ = "0.1.0"
