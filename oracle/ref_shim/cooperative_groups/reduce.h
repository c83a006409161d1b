/* cooperative_groups/reduce.h shim: included by the reference, nothing of it is used on the forward path */
#pragma once
