// forwards to the stand-in types of this repository (include/compat/README.md)
#pragma once
#include <ROFT/CartesianQuaternionModel.h>
