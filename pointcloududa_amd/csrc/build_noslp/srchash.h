#define PCUDA_SRC_HASH "926f76fcbb3e32d6"
