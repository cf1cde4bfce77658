// csmp.hip -- host side of libcsmp.so: the C ABI of include/csmp.h over the gfx950 kernels in
// csmp_kernels.hpp and its sibling kernel headers.  One ctx = one GPU + one HIP stream.  A solve is a chain of
// asynchronous launches with all control state (support, stop flags) in device memory: the host never
// synchronises inside a solve, only when results are copied back.
// The host code is cut by solver family into host/*.hpp, included below IN ORDER (helpers are file-local
// functions defined before their users); this file and csmp_screen.hip are the library's two translation units.
#include "../../include/csmp.h"
#include "../../include/csmp_internal.h"
#include "csmp_kernels.hpp"
#include "csmp_batched.hpp"
#include "csmp_screened.hpp"
#include "csmp_block.hpp"
#include "csmp_forward.hpp"
#include "csmp_downdate.hpp"
#include "csmp_tinv.hpp"
#include "csmp_shard.hpp"
#include "csmp_gram.hpp"
#include "csmp_swap.hpp"

#include <algorithm>
#include <iterator>
#include <iterator>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>  // (std::this_thread::yield in csmp_sp_batch's polling loop; the library starts no threads)
#include <vector>

using namespace csmp;

#include "host/hostonly.hpp"
#include "host/track.hpp"
#include "host/ctx.hpp"
#include "host/lifetime.hpp"
#include "host/dictionary.hpp"
#include "host/chain.hpp"
#include "host/omp.hpp"
#include "host/forward.hpp"
#include "host/steps_sharding.hpp"
#include "host/removal.hpp"
#include "host/gomp_sp.hpp"
#include "host/twostage.hpp"
#include "host/steps_twostage.hpp"
#include "host/batched.hpp"
#include "host/screened.hpp"
#include "host/measure.hpp"
#include "host/rccl.hpp"
