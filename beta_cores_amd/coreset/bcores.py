from .greedy_vi import BetaCoreset   # module name kept for `bayesiancoresets.coreset.bcores` users
