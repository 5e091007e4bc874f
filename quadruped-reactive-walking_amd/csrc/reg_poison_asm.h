// The body of reg_poison_kernel: every architectural and accumulation vector register of the wavefront set to all-ones (a NaN in
// either half of a double), as an assembler loop.  Diagnostic only (QRW_DEBUG_POISON_LDS / qrw_test_known_answer).
#define QRW_REG_POISON_ASM            \
  ".set qrw_poison_i, 0\n\t"          \
  ".rept 256\n\t"                     \
  "v_mov_b32 v[qrw_poison_i], -1\n\t" \
  ".set qrw_poison_i, qrw_poison_i + 1\n\t" \
  ".endr\n\t"                         \
  ".set qrw_poison_i, 0\n\t"          \
  ".rept 256\n\t"                     \
  "v_accvgpr_write_b32 a[qrw_poison_i], v0\n\t" \
  ".set qrw_poison_i, qrw_poison_i + 1\n\t" \
  ".endr\n\t"                         \
  "s_nop 0"
// the clobber list (all 512 registers: it is what makes the compiler give the kernel the whole register file)
#define QRW_R10(p, t) p #t "0", p #t "1", p #t "2", p #t "3", p #t "4", p #t "5", p #t "6", p #t "7", p #t "8", p #t "9"
#define QRW_R256(p)                                                                                                              \
  QRW_R10(p, ), QRW_R10(p, 1), QRW_R10(p, 2), QRW_R10(p, 3), QRW_R10(p, 4), QRW_R10(p, 5), QRW_R10(p, 6), QRW_R10(p, 7),         \
  QRW_R10(p, 8), QRW_R10(p, 9), QRW_R10(p, 10), QRW_R10(p, 11), QRW_R10(p, 12), QRW_R10(p, 13), QRW_R10(p, 14), QRW_R10(p, 15),  \
  QRW_R10(p, 16), QRW_R10(p, 17), QRW_R10(p, 18), QRW_R10(p, 19), QRW_R10(p, 20), QRW_R10(p, 21), QRW_R10(p, 22),                \
  QRW_R10(p, 23), QRW_R10(p, 24), p "250", p "251", p "252", p "253", p "254", p "255"
#define QRW_REG_POISON_CLOBBERS QRW_R256("v"), QRW_R256("a")
