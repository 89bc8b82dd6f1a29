# Identical to the reference's generated shims (R/RcppExports.R:15-17, :34-36): same names,
# formals and defaults, same .Call symbols -- clusterbreak(sim_fn = ...) keeps working unchanged.

similarityMH <- function(sequences, k = 4L, n_hash = 50L) {
    .Call(`_DynaAlign_similarityMH`, sequences, k, n_hash)
}

similarityNW <- function(sequences, matrixName = "BLOSUM62", gapOpen = 10L, gapExt = 4L) {
    .Call(`_DynaAlign_similarityNW`, sequences, matrixName, gapOpen, gapExt)
}
