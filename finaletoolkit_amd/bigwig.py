"""bigWig output for multi_wps (fixedStep sections).  Placeholder until the
writer lands: the kernels and the bedGraph.gz path do not depend on it."""


def write_fixed_step_bigwig(output_file, header, interval_scores):
    raise NotImplementedError("bigWig (.bw) output is not implemented yet; use a .bed.gz / .bedGraph.gz output path")
