// placeholder until client_kernels.hip lands (next commit): every entry fails loudly, nothing falls back to the CPU
#include "client.h"
namespace hydia {
[[noreturn]] static void nyi(const char *w) { throw std::runtime_error(std::string("hydia: ") + w + " is not built in this revision"); }
void client_keygen(Context &, const uint8_t[32]) { nyi("GPU key generation"); }
Ct client_encrypt(Context &, const double *, int, const uint8_t[32], uint64_t) { nyi("GPU encryption"); }
Ct client_encrypt_query(Context &, const double *, const uint8_t[32], uint64_t) { nyi("GPU query encryption"); }
void client_decrypt(Context &, const Ct &, double *) { nyi("GPU decryption"); }
void client_enroll(Context &, double *, size_t, const uint8_t[32]) { nyi("GPU enrolment"); }
}  // namespace hydia
