"""Task-id validation (reference: ``loco_mujoco/utils/checks.py:3-76``)."""


def check_validity_task_mode_dataset(env_name, task, mode, dataset_type,
                                     valid_tasks, valid_modes, valid_dataset_types, non_combinable):
    def _check(kind, value, valid):
        if valid is None:
            if value is not None:
                raise ValueError("%s does not take a %s, but '%s' was given." % (env_name, kind, value))
        elif value not in valid:
            raise ValueError("'%s' is not a valid %s for %s. Valid: %s." % (value, kind, env_name, valid))

    _check("task", task, valid_tasks)
    _check("mode", mode, valid_modes)
    _check("dataset type", dataset_type, valid_dataset_types)
    for bad_t, bad_m, bad_dt in (non_combinable or []):
        if ((bad_t is None or task == bad_t) and (bad_m is None or mode == bad_m)
                and (bad_dt is None or dataset_type == bad_dt)):
            raise ValueError("The combination task=%s, mode=%s, dataset_type=%s is not available for %s."
                             % (task, mode, dataset_type, env_name))
