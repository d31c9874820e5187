// libowshen_verify.so: og_verify + og_last_error and nothing else, for hosts WITHOUT ROCm user space.
//
// The `burn_tx` seam (/root/reference/src/blockchain/tx/burn_tx.rs:11-32) runs on every node that replays blocks; such a
// node needs the Groth16 check, not the prover.  libowshen_gpu.so links libamdhip64 and librccl, so dlopen-ing it just to
// verify a proof drags the whole ROCm user space in.  This translation unit plus verify.hip compiled for
// the host only (`--offload-host-only`: same field / curve / pairing source, no device code, no HIP runtime call) link
// into a library whose only dependencies are libstdc++ and libc.  Same entry points, same bytes in and out as the full
// library's og_verify (include/owshen_gpu.h).
#include "ctx.h"

namespace og {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int verify_cpu(const uint8_t*, size_t, const uint8_t*, size_t, const uint8_t*, int*);

}  // namespace og

extern "C" {

const char* og_last_error(void) { return og::g_err.c_str(); }

int og_verify(const uint8_t* vk, size_t vk_len, const uint8_t* public_inputs, size_t n_pub, const uint8_t proof[256], int* ok_out) {
  return og::guarded([&]() -> int {
    OG_REQUIRE(vk != nullptr && proof != nullptr && ok_out != nullptr && (n_pub == 0 || public_inputs != nullptr),
               "og_verify: null argument");
    return og::verify_cpu(vk, vk_len, public_inputs, n_pub, proof, ok_out);
  });
}

}  // extern "C"
