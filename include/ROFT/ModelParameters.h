// ROFT/ModelParameters.h -- the reference's header name; the class lives in Sources.h (see the file:line references there).
#pragma once
#include "Sources.h"
