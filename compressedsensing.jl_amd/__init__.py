"""compressedsensing.jl_amd -- MI355X-native matching pursuit (mp / omp / gomp / sp).

Host-side mirror of CompressedSensing.jl's matching-pursuit interface over libcsmp.so
(hand-written gfx950 HIP kernels behind the C ABI of include/csmp.h).  The directory name
contains a dot (layout contract), so load this package with `csmp_pkg.load()`.
"""
from .sparsevec import SparseVector, spzeros
from .data import sparse_vector, sparse_data, gaussian_data, perturb, samesupport, structured_dictionary
from ._lib import CsmpError, Context, LIB_PATH, comm_id, write_dictionary_file, dictionary_file_info
from .api import (Dictionary, mp, omp, gomp, sp, ompr, srr, rmp, foba, br, fbr, lace, fr, ols, oomp, ormp, FR, OLS, omp_batch, omp_batch_mfma, gomp_batch, sp_batch, solve_in_flight, fr_batch, MP, OMP, GOMP, SP, SubspacePursuit, OMPR, sp_acquisition, update_, argmaxinner,
                  oblivious, oblivious_acquisition, random_acquisition)
from .sharded import omp_sharded, shard_range, sharded_solve, fr_sharded, omp_colsharded, HipColumnShard, column_range, pack_t, gather_packed, unpack_t, library_comm, ShardedSolveError, ShardedArgumentError

__all__ = [
    "SparseVector", "spzeros", "sparse_vector", "sparse_data", "gaussian_data", "perturb", "samesupport", "structured_dictionary",
    "CsmpError", "Context", "Dictionary", "mp", "omp", "gomp", "sp", "ompr", "srr", "rmp", "foba", "br", "fbr", "lace", "fr", "ols", "oomp", "ormp", "FR", "OLS", "omp_batch", "omp_batch_mfma", "gomp_batch", "sp_batch", "solve_in_flight", "fr_batch", "MP", "OMP", "GOMP", "SP", "SubspacePursuit", "OMPR", "sp_acquisition",
    "oblivious", "oblivious_acquisition", "random_acquisition",
    "update_", "argmaxinner", "omp_sharded", "shard_range", "sharded_solve", "fr_sharded", "omp_colsharded", "HipColumnShard", "column_range",
    "pack_t", "gather_packed", "unpack_t",
]
