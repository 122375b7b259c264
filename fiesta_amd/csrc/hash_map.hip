// placeholder translation unit; the hash-block map is implemented later in this round
#include "hash_map.hpp"
