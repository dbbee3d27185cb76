// main.h — the include the reference's src/main.cpp asks for; see sff_dropin.h.
#pragma once
#include "sff_dropin.h"
