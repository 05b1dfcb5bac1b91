#!/bin/bash
# parse_options.sh -- to be SOURCED by a recipe ( ". parse_options.sh || exit 1;", egs/*/run.sh ) after it has set the
# defaults of its option variables.  Written for this repository (the recipes find it on PATH through path.sh, where the
# reference keeps a script of the same name and calling convention).
#
#   --some-name VALUE   sets the shell variable some_name (dashes become underscores).  The variable must already
#                       exist -- an unknown option is an error, which catches typos -- and a variable whose default
#                       is "true" or "false" only accepts one of these two words.
#   --config FILE       FILE is sourced BEFORE the other options are applied, so the command line wins.
#   --help | -h         prints $help_message (if the recipe defined one) and exits.
#   --                  ends option parsing; so does the first argument that does not start with "--".
# The recipe's positional parameters are left holding what follows the options.

# pass 1: config files, left to right
__po_n=$#
for ((__po_i = 1; __po_i < __po_n; __po_i++)); do
  if [ "${!__po_i}" = "--config" ]; then
    __po_j=$((__po_i + 1))
    __po_file="${!__po_j}"
    if [ ! -r "${__po_file}" ]; then
      echo "$0: cannot read the config file '${__po_file}'" 1>&2
      exit 1
    fi
    . "${__po_file}"
  fi
done

# pass 2: the options themselves
while [ $# -gt 0 ]; do
  case "$1" in
    --help | -h)
      if [ -n "${help_message:-}" ]; then printf '%s\n' "${help_message}" 1>&2; else echo "$0: no help available." 1>&2; fi
      exit 0
      ;;
    --)
      shift
      break
      ;;
    --*=*)
      echo "$0: options are given as '--name value', not '$1'" 1>&2
      exit 1
      ;;
    --*)
      __po_name="${1#--}"
      __po_name="${__po_name//-/_}"
      if [ $# -lt 2 ]; then
        echo "$0: option $1 needs a value" 1>&2
        exit 1
      fi
      if [ "${__po_name}" != "config" ]; then
        if ! [[ "${__po_name}" =~ ^[A-Za-z_][A-Za-z0-9_]*$ ]] || [ -z "${!__po_name+set}" ]; then
          echo "$0: invalid option $1" 1>&2
          exit 1
        fi
        __po_old="${!__po_name}"
        if { [ "${__po_old}" = "true" ] || [ "${__po_old}" = "false" ]; } && [ "$2" != "true" ] && [ "$2" != "false" ]; then
          echo "$0: option $1 expects true or false, got '$2'" 1>&2
          exit 1
        fi
        printf -v "${__po_name}" '%s' "$2"
      elif [ -n "${config+set}" ]; then
        # a recipe that has its own variable called "config" (egs/*/run.sh: the model configuration path) gets it set too
        printf -v config '%s' "$2"
      fi
      shift 2
      ;;
    *)
      break
      ;;
  esac
done
unset __po_n __po_i __po_j __po_file __po_name __po_old
true
