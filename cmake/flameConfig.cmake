# cmake/flameConfig.cmake -- lets flame_ros' `find_package(flame REQUIRED)` (reference
# CMakeLists.txt:57) resolve to this repository: it defines the two variables the reference build
# consumes, flame_INCLUDE_DIRS (reference CMakeLists.txt:205 include_directories) and
# flame_LIBRARIES (reference src/CMakeLists.txt:11,34,58 target_link_libraries).
#
#   cmake -Dflame_DIR=<repo>/cmake ...        or        list(APPEND CMAKE_PREFIX_PATH <repo>)
#
# include/flame/*.h is header-only; the one binary is flame_ros_amd/libflame_hip.so (HIP kernels +
# C ABI), built by `python -c "import __graft_entry__ as g; g.build()"`.
get_filename_component(_flame_root "${CMAKE_CURRENT_LIST_DIR}/.." ABSOLUTE)
set(flame_INCLUDE_DIRS "${_flame_root}/include")
find_library(flame_HIP_LIBRARY NAMES flame_hip PATHS "${_flame_root}/flame_ros_amd" NO_DEFAULT_PATH)
if(NOT flame_HIP_LIBRARY)
  set(flame_FOUND FALSE)
  if(flame_FIND_REQUIRED)
    message(FATAL_ERROR "flame: ${_flame_root}/flame_ros_amd/libflame_hip.so not found -- build it first "
                        "(python -c \"import __graft_entry__ as g; g.build()\")")
  endif()
else()
  find_package(Threads REQUIRED)
  set(flame_LIBRARIES "${flame_HIP_LIBRARY}" Threads::Threads)
  set(flame_FOUND TRUE)
  if(NOT TARGET flame::flame)
    add_library(flame::flame INTERFACE IMPORTED)
    set_target_properties(flame::flame PROPERTIES
      INTERFACE_INCLUDE_DIRECTORIES "${flame_INCLUDE_DIRS}"
      INTERFACE_LINK_LIBRARIES "${flame_LIBRARIES}")
  endif()
endif()
unset(_flame_root)
