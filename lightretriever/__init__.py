"""Import-path shim: `lightretriever.*` names that eval/eval_utils.py and eval/evaluate_mteb.py of the reference import
("Module surface required by eval/"), served by lightretriever_amd.  Dense asymmetric path only."""
