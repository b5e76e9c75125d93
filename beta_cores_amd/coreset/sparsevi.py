from .greedy_vi import SparseVICoreset   # module name kept for `bayesiancoresets.coreset.sparsevi` users
