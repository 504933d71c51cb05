// The 64-dof build of the general-tree engine (csrc/so101_tree.hpp, TREE_VARIANT 64: the Dining scenes): the same source as
// tu_tree.hip compiled with the larger limits inside namespace tv64, entry points so101_tree64_*.
#define TREE_VARIANT 64
#include "tu_tree.hip"
