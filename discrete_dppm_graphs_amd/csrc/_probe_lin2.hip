#include "dppm_wsplit.h"
namespace dppm { int lin2_probe(const WsArgs& a, hipStream_t st) { return w3_lin2<3>(a, st); } }
